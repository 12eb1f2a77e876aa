// rust/model.rs — hands skani's embedded learned-ANI model to libpyskani_amd.so (new file src/pyskani/_skani/model.rs).
//
// Replaces nothing in the reference by itself: it is what lets the `&model_opt` argument of
// skani::chain::map_params_from_sketch (src/pyskani/_skani/lib.rs:646-651) cross the C-ABI. The weights come out of
// skani::regression::get_model (lib.rs:614) — they are compiled into the skani crate, which this repository cannot read —
// and are flattened into the PskTreeNode array psk_model_create takes (include/pyskani_amd.h, "learned-ANI regression").
//
// NOT compiled in this repository (no Rust toolchain in the build image). Written against gbdt 0.1.3 (Cargo.lock:1608):
// `GBDT { conf: Config, trees: Vec<DecisionTree>, bias: ValueType, .. }`, `DecisionTree { tree: BinaryTree<DTNode>, .. }`,
// `DTNode { feature_index, feature_value, pred, missing, is_leaf, .. }`, children through `BinaryTree::get_left_child /
// get_right_child`. Those fields are private in gbdt 0.1.3, so the flattening goes through the crate's own serde form
// (`serde_json::to_value(&gbdt)`), which is exactly the JSON psk_model_load_json parses on the C side; a maintainer with a
// fork that exposes the fields can build the node array directly with `flatten_tree` below.
use crate::ffi::{self, PskCtx, PskModel, PskTreeNode};
use std::os::raw::c_char;

/// One flattened tree: nodes in depth-first order, child links relative to the tree's first node.
pub struct FlatModel {
    pub nodes: Vec<PskTreeNode>,
    pub tree_first_node: Vec<u32>,
    pub bias: f32,
    pub shrinkage: f32,
}

/// `gbdt::GBDT` -> flat arrays, through the crate's serde representation (tree = {"tree": {"tree": [nodes], "root": i}}
/// where every node is {"value": DTNode, "index", "left", "right"}; 0 = no child).
pub fn to_psk_model(g: &gbdt::gradient_boost::GBDT) -> Result<FlatModel, String> {
    let v = serde_json::to_value(g).map_err(|e| e.to_string())?;
    let conf = &v["conf"];
    let shrinkage = conf["shrinkage"].as_f64().ok_or("conf.shrinkage")? as f32;
    let bias = v["bias"].as_f64().unwrap_or(0.0) as f32;
    let mut nodes = Vec::new();
    let mut first = Vec::new();
    for t in v["trees"].as_array().ok_or("trees")? {
        let arena = t["tree"]["tree"].as_array().ok_or("tree.tree")?;
        let root = t["tree"]["root"].as_u64().unwrap_or(0) as usize;
        first.push(nodes.len() as u32);
        flatten_tree(arena, root, &mut nodes)?;
    }
    Ok(FlatModel { nodes, tree_first_node: first, bias, shrinkage })
}

/// Depth-first copy of one tree's arena starting at `root`; returns the index (relative to the tree) of the copied node.
fn flatten_tree(arena: &[serde_json::Value], root: usize, out: &mut Vec<PskTreeNode>) -> Result<i32, String> {
    let base = out.len();
    // explicit stack: (arena index, slot of the parent's link to patch, is_right)
    let mut stack = vec![(root, usize::MAX, false)];
    while let Some((ai, parent, is_right)) = stack.pop() {
        let n = arena.get(ai).ok_or("node index out of range")?;
        let d = &n["value"];
        let me = out.len();
        out.push(PskTreeNode {
            feature: d["feature_index"].as_i64().unwrap_or(0) as i32,
            threshold: d["feature_value"].as_f64().unwrap_or(0.0) as f32,
            left: -1,
            right: -1,
            value: d["pred"].as_f64().unwrap_or(0.0) as f32,
            missing: d["missing"].as_i64().unwrap_or(0) as i32,
            is_leaf: d["is_leaf"].as_bool().unwrap_or(false) as i32,
            reserved: 0,
        });
        if parent != usize::MAX {
            let rel = (me - base) as i32;
            if is_right { out[parent].right = rel } else { out[parent].left = rel }
        }
        let (l, r) = (n["left"].as_u64().unwrap_or(0) as usize, n["right"].as_u64().unwrap_or(0) as usize);
        if r != 0 { stack.push((r, me, true)); }
        if l != 0 { stack.push((l, me, false)); }
    }
    Ok(0)
}

/// The model `Database::query` needs for (c, learned): built once and cached by the caller (skani re-parses its embedded JSON
/// on every call, lib.rs:614 — 100 000 short-contig queries pay for that 100 000 times).
pub unsafe fn upload(ctx: *mut PskCtx, c: usize, learned: bool) -> pyo3::PyResult<*mut PskModel> {
    let mut out: *mut PskModel = std::ptr::null_mut();
    match skani::regression::get_model(c, learned) {
        None => Ok(out),      // `model_opt == None`: PskQueryOpts.model stays null, psk_query returns the raw chain ANI
        Some(g) => {
            // feature order: the `psk_feature` default of include/pyskani_amd.h — CHECK it against
            // skani::regression::predict_from_ani_res before trusting learned results (this repository could not read that function)
            let m = to_psk_model(&g).map_err(pyo3::exceptions::PyRuntimeError::new_err)?;
            ffi::check(ffi::psk_model_create(
                ctx, m.nodes.as_ptr(), m.nodes.len() as u64, m.tree_first_node.as_ptr(), m.tree_first_node.len() as u32,
                m.bias, m.shrinkage, std::ptr::null(), 0, &mut out,
            ))?;
            Ok(out)
        }
    }
}

/// The alternative route: let the C side parse the crate's own JSON (same semantics, no Rust-side flattening).
pub unsafe fn upload_json(ctx: *mut PskCtx, g: &gbdt::gradient_boost::GBDT) -> pyo3::PyResult<*mut PskModel> {
    let text = serde_json::to_string(g).map_err(|e| pyo3::exceptions::PyRuntimeError::new_err(e.to_string()))?;
    let mut out: *mut PskModel = std::ptr::null_mut();
    ffi::check(ffi::psk_model_load_json(ctx, text.as_ptr() as *const c_char, text.len(), &mut out))?;
    Ok(out)
}
