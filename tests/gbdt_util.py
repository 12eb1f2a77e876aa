"""Synthetic gradient-boosted regression trees for the learned-ANI tests: a generator, a 20-line reference evaluator
(gbdt 0.1.3 semantics, f32) and a writer of the serde-JSON shape of gbdt::gradient_boost::GBDT."""
import json

import numpy as np

UNKNOWN = np.float32(-3.402823466e+38)     # gbdt VALUE_TYPE_UNKNOWN = f32::MIN
FEATURES = ["ani100", "std100", "q90_query", "q50_query", "q10_query", "q90_ref", "q50_ref", "q10_ref", "avg_chain_len"]


def random_trees(rng, n_trees=3, depth=3, n_features=9, scales=None):
    """Full binary trees as node lists [(feature, threshold, left, right, value, missing, is_leaf)], root = node 0."""
    scales = scales or [(97.0, 100.0), (0.0, 1.0)] + [(1e3, 6e6)] * 6 + [(1e3, 3e4)]
    trees = []
    for _ in range(n_trees):
        nodes, n_inner = [], 2 ** depth - 1
        for i in range(2 ** (depth + 1) - 1):
            leaf = i >= n_inner
            f = int(rng.integers(0, n_features))
            lo, hi = scales[f]
            nodes.append((f, float(np.float32(rng.uniform(lo, hi))), 0 if leaf else 2 * i + 1, 0 if leaf else 2 * i + 2,
                          float(np.float32(rng.uniform(-0.3, 0.3))), int(rng.integers(-1, 2)), int(leaf)))
        trees.append(nodes)
    return trees


def predict(trees, bias, shrinkage, row):
    """gbdt 0.1.3 GBDT::predict (SquaredError) for one feature row: f32 accumulation in tree order."""
    acc = np.float32(bias)
    for nodes in trees:
        i = 0
        while True:
            f, thr, left, right, value, missing, leaf = nodes[i]
            if leaf:
                break
            x = np.float32(row[f])
            go = missing if x == UNKNOWN else (-1 if x < np.float32(thr) else 1)
            if go == 0:
                break
            i = left if go < 0 else right
        acc = np.float32(acc + np.float32(shrinkage) * np.float32(value))
    return float(acc)


def to_gbdt_json(trees, bias, shrinkage, features=None):
    doc = {"conf": {"feature_size": len(features or FEATURES), "max_depth": 8, "iterations": len(trees), "shrinkage": shrinkage,
                    "feature_sample_ratio": 1.0, "data_sample_ratio": 1.0, "min_leaf_size": 1, "loss": "SquaredError",
                    "debug": False, "initial_guess_enabled": False, "training_optimization_level": 2},
           "trees": [{"tree": {"tree": [{"value": {"feature_index": f, "feature_value": thr, "pred": val, "missing": mis, "is_leaf": bool(leaf)},
                                          "index": i, "left": l, "right": r} for i, (f, thr, l, r, val, mis, leaf) in enumerate(t)]},
                      "feature_size": len(features or FEATURES), "max_depth": 8, "min_leaf_size": 1, "loss": "SquaredError",
                      "feature_sample_ratio": 1.0} for t in trees],
           "bias": bias}
    if features is not None:
        doc["psk_features"] = list(features)
    return json.dumps(doc)
