#!/bin/bash
# A/B builds: tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> ...  ->  pyskani_amd/libpyskani_amd_<name>.so (the named files recompiled with the flags,
# every other object as built by `make`); run with PSK_LIB_PATH=$PWD/pyskani_amd/libpyskani_amd_<name>.so
set -e
cd "$(dirname "$0")/../pyskani_amd/csrc"
name=$1; flags=$2; shift 2
mkdir -p /tmp/variant_$name
objs=""
for o in capi sketch screen join dp select reduce chain query_many seed_index slice_join small_query model exchange pack_host; do
  use=$o.o
  for f in "$@"; do if [ "$f" = "$o.hip" ]; then /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $flags -c $f -o /tmp/variant_$name/$o.o & use=/tmp/variant_$name/$o.o; fi; done
  objs="$objs $use"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpyskani_amd_$name.so $objs -ldl
ls -la ../libpyskani_amd_$name.so
