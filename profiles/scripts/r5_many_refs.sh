#!/bin/bash
# round 5: the cost of the 65 536-reference cliff: the many-references workload of tests/test_gpu_scale_paths.py (2 000 rescued contigs against N references of 2-2.8 kb)
# at N = 60 000 (seed indexes) and N = 70 000 (probe-table join + per-reference prefilter); prints: hits, digest, device bytes kept, seconds of the second query_many
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 60000 70000; do
  echo "== N = $n"
  PSK_TEST_MANY_REFS=$n PSK_TEST_SAMPLE=/tmp/sample_$n.pkl timeout 900 python3 - <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import test_gpu_scale_paths as T
exec(compile("import os\n" + T.MANY_REFS, "many_refs", "exec"))
PY
done
