// Learned-ANI regression: skani::regression::{use_learned_ani, get_model} at
// /root/reference/src/pyskani/_skani/lib.rs:611-614 and the `&model_opt` handed to map_params_from_sketch at :646-651.
// skani embeds trained gradient-boosted trees (crate gbdt 0.1.3, Cargo.lock:1608) in its own source, which is NOT part
// of the reference tree, so the weights are user supplied: flat arrays (psk_model_create) or the serde-JSON text of a
// gbdt::gradient_boost::GBDT (psk_model_load_json / _file). Inference is a HIP kernel, one lane per hit.
#include "common.h"
#include <cmath>
#include <fstream>
#include <map>
#include <sstream>

// ------------------------------------------------------------------ device evaluation
// gbdt 0.1.3 DecisionTree::predict_one + GBDT::predict (SquaredError): f32 throughout, trees accumulated in order.
template <class F>
__device__ __forceinline__ float gbdt_eval(const ModelDev& M, F feat) {
    float acc = M.bias;
    for (uint32_t t = 0; t < M.n_trees; t++) {
        const ModelNode* __restrict__ T = M.nodes + M.first[t];
        const uint32_t tn = M.first[t + 1] - M.first[t];
        uint32_t i = 0;
        float v = 0.f;
        for (uint32_t step = 0; step <= tn; step++) {      // a well-formed tree ends long before tn steps
            const ModelNode nd = T[i];
            v = nd.value;
            if (nd.is_leaf) break;
            const float x = feat(nd);
            int go;                                          // -1 left, 0 stop, +1 right
            if (x == PSK_FEATURE_UNKNOWN) go = nd.missing;
            else go = x < nd.threshold ? -1 : 1;
            if (go == 0) break;
            i = (uint32_t)(go < 0 ? nd.left : nd.right);
        }
        acc += M.shrinkage * v;
    }
    return acc;
}

__global__ __launch_bounds__(256) void model_predict_kernel(ModelDev M, const float* __restrict__ rows, uint32_t n_rows, float* __restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const float* x = rows + (size_t)r * M.n_features;
    out[r] = gbdt_eval(M, [&](const ModelNode& nd) { return x[nd.feature]; });
}

// one lane per hit: assemble the feature menu in LDS, evaluate, overwrite ani (kept in ani_raw)
__global__ __launch_bounds__(128) void learned_ani_kernel(ModelDev M, psk_hit* __restrict__ hits, const uint2* __restrict__ pair_qr,
                                                          const SketchDesc* __restrict__ qd, const SketchDesc* __restrict__ rd, uint32_t n_pairs) {
    __shared__ float s_f[128][PSK_F_COUNT + 1];
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    psk_hit h = hits[p];
    if (!(h.ani > 0.f)) return;            // invalid result (aligned fraction below the cut-off): left as it is
    float* f = s_f[threadIdx.x];
    const SketchDesc& Q = qd[pair_qr[p].x]; const SketchDesc& R = rd[pair_qr[p].y];
    f[PSK_F_ANI100] = h.ani_raw * 100.f; f[PSK_F_STD100] = h.ani_std * 100.f;
    f[PSK_F_Q90_QUERY] = Q.lenq[0]; f[PSK_F_Q50_QUERY] = Q.lenq[1]; f[PSK_F_Q10_QUERY] = Q.lenq[2];
    f[PSK_F_Q90_REF] = R.lenq[0]; f[PSK_F_Q50_REF] = R.lenq[1]; f[PSK_F_Q10_REF] = R.lenq[2];
    f[PSK_F_AVG_CHAIN_LEN] = h.n_intervals ? (float)h.covered_query / (float)h.n_intervals : 0.f;
    f[PSK_F_AF_QUERY] = h.af_query; f[PSK_F_AF_REF] = h.af_ref; f[PSK_F_N_CHUNKS] = (float)h.n_chunks;
    f[PSK_F_TOTAL_LEN_QUERY] = (float)Q.total_len; f[PSK_F_TOTAL_LEN_REF] = (float)R.total_len;
    f[PSK_F_N_CONTIGS_QUERY] = (float)Q.n_contigs; f[PSK_F_N_CONTIGS_REF] = (float)R.n_contigs;
    float pred = gbdt_eval(M, [&](const ModelNode& nd) { return f[nd.menu]; }) * 0.01f;
    pred = pred < 0.f ? 0.f : (pred > 1.f ? 1.f : pred);
    h.ani = pred; h.learned = 1;
    hits[p] = h;
}

void learned_apply_launch(const psk_model* m, psk_hit* d_hits, const uint2* pair_qr, const SketchDesc* qd, const SketchDesc* rd, uint32_t n_pairs, hipStream_t st) {
    if (!m || !n_pairs) return;
    hipLaunchKernelGGL(learned_ani_kernel, dim3((n_pairs + 127) / 128), dim3(128), 0, st, m->dev, d_hits, pair_qr, qd, rd, n_pairs);
}

// ------------------------------------------------------------------ minimal JSON reader (objects, arrays, numbers, strings, literals)
namespace {
struct JVal {
    enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    double num = 0; bool b = false; std::string str;
    std::vector<JVal> arr; std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* key) const { for (auto& kv : obj) if (kv.first == key) return &kv.second; return nullptr; }
};
struct JParser {
    const char* p; const char* e; std::string err; int depth = 0;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool str(std::string& out) {
        if (p >= e || *p != '"') return fail("expected string");
        p++;
        while (p < e && *p != '"') {
            if (*p == '\\') {
                if (++p >= e) return fail("bad escape");
                switch (*p) { case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': if (e - p < 5) return fail("bad \\u"); out += '?'; p += 4; break;
                    default: out += *p; }
                p++;
            } else out += *p++;
        }
        if (p >= e) return fail("unterminated string");
        p++;
        return true;
    }
    bool val(JVal& v) {
        if (++depth > 64) return fail("nesting too deep");
        ws();
        if (p >= e) return fail("unexpected end");
        bool ok = true;
        if (*p == '{') {
            v.kind = JVal::OBJ; p++; ws();
            if (p < e && *p == '}') p++;
            else for (;;) {
                ws(); std::string k; if (!str(k)) { ok = false; break; }
                ws(); if (p >= e || *p != ':') { ok = fail("expected ':'"); break; } p++;
                v.obj.emplace_back(k, JVal()); if (!val(v.obj.back().second)) { ok = false; break; }
                ws(); if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == '}') { p++; break; }
                ok = fail("expected ',' or '}'"); break;
            }
        } else if (*p == '[') {
            v.kind = JVal::ARR; p++; ws();
            if (p < e && *p == ']') p++;
            else for (;;) {
                v.arr.emplace_back(); if (!val(v.arr.back())) { ok = false; break; }
                ws(); if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; break; }
                ok = fail("expected ',' or ']'"); break;
            }
        } else if (*p == '"') { v.kind = JVal::STR; ok = str(v.str); }
        else if (e - p >= 4 && !strncmp(p, "true", 4)) { v.kind = JVal::BOOL; v.b = true; p += 4; }
        else if (e - p >= 5 && !strncmp(p, "false", 5)) { v.kind = JVal::BOOL; v.b = false; p += 5; }
        else if (e - p >= 4 && !strncmp(p, "null", 4)) { v.kind = JVal::NUL; p += 4; }
        else {
            char* end = nullptr;
            std::string tmp(p, (size_t)std::min<ptrdiff_t>(e - p, 64));
            v.num = strtod(tmp.c_str(), &end);
            if (end == tmp.c_str()) ok = fail("unexpected character");
            else { v.kind = JVal::NUM; p += end - tmp.c_str(); }
        }
        depth--;
        return ok;
    }
};
const char* const FEATURE_NAMES[PSK_F_COUNT] = {"ani100", "std100", "q90_query", "q50_query", "q10_query", "q90_ref", "q50_ref", "q10_ref",
                                                "avg_chain_len", "af_query", "af_ref", "n_chunks", "total_len_query", "total_len_ref",
                                                "n_contigs_query", "n_contigs_ref"};
const int32_t DEFAULT_FEATURES[9] = {PSK_F_ANI100, PSK_F_STD100, PSK_F_Q90_QUERY, PSK_F_Q50_QUERY, PSK_F_Q10_QUERY, PSK_F_Q90_REF, PSK_F_Q50_REF,
                                     PSK_F_Q10_REF, PSK_F_AVG_CHAIN_LEN};
double jnum(const JVal* v, double dflt) { return v && v->kind == JVal::NUM ? v->num : (v && v->kind == JVal::BOOL ? (v->b ? 1.0 : 0.0) : dflt); }
}  // namespace

extern "C" {

psk_status psk_model_create(psk_ctx* ctx, const psk_tree_node* nodes, uint64_t n_nodes, const uint32_t* tree_first_node, uint32_t n_trees,
                            float bias, float shrinkage, const int32_t* features, uint32_t n_features, psk_model** out) {
    if (!ctx || !out || (n_nodes && !nodes) || (n_trees && !tree_first_node)) { psk_set_error("model_create: NULL argument"); return PSK_EINVAL; }
    *out = nullptr;
    if (!features) { features = DEFAULT_FEATURES; n_features = 9; }
    if (n_features == 0 || n_features > 64) { psk_set_error("model_create: a model needs 1..64 features"); return PSK_EINVAL; }
    if (n_nodes >= (1ull << 31)) { psk_set_error("model_create: too many nodes"); return PSK_ELIMIT; }
    for (uint32_t j = 0; j < n_features; j++) if (features[j] < 0 || features[j] >= PSK_F_COUNT) { psk_set_error("model_create: unknown feature id %d", features[j]); return PSK_EINVAL; }
    std::vector<ModelNode> h(n_nodes ? n_nodes : 1);
    std::vector<uint32_t> first(n_trees + 1);
    for (uint32_t t = 0; t < n_trees; t++) {
        const uint64_t a = tree_first_node[t], b = t + 1 < n_trees ? tree_first_node[t + 1] : n_nodes;
        if (a >= b || b > n_nodes) { psk_set_error("model_create: tree %u is empty or out of range", t); return PSK_EINVAL; }
        first[t] = (uint32_t)a;
        for (uint64_t i = a; i < b; i++) {
            const psk_tree_node& s = nodes[i];
            ModelNode& d = h[i];
            d.is_leaf = s.is_leaf != 0; d.value = s.value; d.threshold = s.threshold; d.missing = s.missing < 0 ? -1 : (s.missing > 0 ? 1 : 0);
            d.feature = 0; d.menu = 0; d.left = d.right = 0;
            if (!d.is_leaf) {
                if (s.feature < 0 || (uint32_t)s.feature >= n_features) { psk_set_error("model_create: node %llu tests feature %d of %u", (unsigned long long)i, s.feature, n_features); return PSK_EINVAL; }
                if (s.left < 0 || s.right < 0 || (uint64_t)s.left >= b - a || (uint64_t)s.right >= b - a) { psk_set_error("model_create: node %llu has a child outside its tree", (unsigned long long)i); return PSK_EINVAL; }
                d.feature = s.feature; d.menu = features[s.feature]; d.left = s.left; d.right = s.right;
            }
        }
    }
    first[n_trees] = (uint32_t)n_nodes;
    std::unique_ptr<psk_model> m(new psk_model());
    m->ctx = ctx; m->n_nodes = n_nodes; m->features.assign(features, features + n_features);
    PSK_HIP(hipSetDevice(ctx->device));
    const size_t nb = sizeof(ModelNode) * h.size(), fb = sizeof(uint32_t) * first.size();
    PSK_HIP(hipMalloc(&m->base, nb + fb + 256));
    PSK_HIP(hipMemcpy(m->base, h.data(), nb, hipMemcpyHostToDevice));
    PSK_HIP(hipMemcpy((char*)m->base + nb, first.data(), fb, hipMemcpyHostToDevice));
    m->dev.nodes = (const ModelNode*)m->base; m->dev.first = (const uint32_t*)((char*)m->base + nb);
    m->dev.n_trees = n_trees; m->dev.n_features = n_features; m->dev.bias = bias; m->dev.shrinkage = shrinkage;
    *out = m.release();
    return PSK_OK;
}

/* serde-JSON of gbdt::gradient_boost::GBDT (gbdt 0.1.3): {"conf": {"shrinkage", "iterations", "feature_size", "loss", "initial_guess_enabled", ...},
 * "trees": [{"tree": {"tree": [{"value": {"feature_index", "feature_value", "pred", "missing", "is_leaf"}, "index", "left", "right"}, ...]}, ...}],
 * "bias"}; node 0 is a tree's root, left/right are indices into the tree's own node array. Optional top-level
 * "psk_features": [names] fixes what each position of the feature vector means. */
psk_status psk_model_load_json(psk_ctx* ctx, const char* json, size_t len, psk_model** out) {
    if (!ctx || !json || !out) { psk_set_error("model_load: NULL argument"); return PSK_EINVAL; }
    *out = nullptr;
    JParser P{json, json + len, "", 0};
    JVal root;
    if (!P.val(root) || root.kind != JVal::OBJ) { psk_set_error("model_load: not a JSON object (%s at byte %zu)", P.err.c_str(), (size_t)(P.p - json)); return PSK_EINVAL; }
    const JVal* conf = root.get("conf"); const JVal* trees = root.get("trees");
    if (!conf || conf->kind != JVal::OBJ || !trees || trees->kind != JVal::ARR) { psk_set_error("model_load: expected the serde-JSON of a gbdt GBDT (\"conf\", \"trees\", \"bias\")"); return PSK_EINVAL; }
    if (const JVal* loss = conf->get("loss")) {
        const std::string l = loss->kind == JVal::STR ? loss->str : (loss->kind == JVal::OBJ && !loss->obj.empty() ? loss->obj[0].first : "");
        if (!l.empty() && l != "SquaredError" && l != "LAD") { psk_set_error("model_load: loss '%s' is not a regression loss this build evaluates", l.c_str()); return PSK_EINVAL; }
    }
    if (jnum(conf->get("initial_guess_enabled"), 0) != 0) { psk_set_error("model_load: models with initial_guess_enabled are not supported"); return PSK_EINVAL; }
    const float shrink = (float)jnum(conf->get("shrinkage"), 1.0), bias = (float)jnum(root.get("bias"), 0.0);
    size_t iters = (size_t)jnum(conf->get("iterations"), (double)trees->arr.size());
    if (iters > trees->arr.size()) iters = trees->arr.size();
    std::vector<int32_t> feats;
    if (const JVal* pf = root.get("psk_features")) {
        if (pf->kind != JVal::ARR) { psk_set_error("model_load: psk_features must be an array of names"); return PSK_EINVAL; }
        for (const JVal& n : pf->arr) {
            int id = -1;
            for (int j = 0; j < PSK_F_COUNT; j++) if (n.kind == JVal::STR && n.str == FEATURE_NAMES[j]) id = j;
            if (id < 0) { psk_set_error("model_load: unknown feature name '%s'", n.kind == JVal::STR ? n.str.c_str() : "?"); return PSK_EINVAL; }
            feats.push_back(id);
        }
    } else {
        feats.assign(DEFAULT_FEATURES, DEFAULT_FEATURES + 9);
        const size_t fs = (size_t)jnum(conf->get("feature_size"), 9);
        if (fs != feats.size()) { psk_set_error("model_load: the model has %zu features but no \"psk_features\" list; the default vector has 9", fs); return PSK_EINVAL; }
    }
    std::vector<psk_tree_node> nodes; std::vector<uint32_t> first;
    for (size_t t = 0; t < iters; t++) {
        const JVal* tt = trees->arr[t].get("tree"); const JVal* arr = tt ? tt->get("tree") : nullptr;
        if (!arr || arr->kind != JVal::ARR || arr->arr.empty()) { psk_set_error("model_load: tree %zu has no node array", t); return PSK_EINVAL; }
        first.push_back((uint32_t)nodes.size());
        for (const JVal& n : arr->arr) {
            const JVal* v = n.get("value");
            if (!v || v->kind != JVal::OBJ) { psk_set_error("model_load: tree %zu: node without \"value\"", t); return PSK_EINVAL; }
            psk_tree_node nd{};
            nd.feature = (int32_t)jnum(v->get("feature_index"), 0); nd.threshold = (float)jnum(v->get("feature_value"), 0);
            nd.value = (float)jnum(v->get("pred"), 0); nd.missing = (int32_t)jnum(v->get("missing"), 0); nd.is_leaf = jnum(v->get("is_leaf"), 0) != 0;
            nd.left = (int32_t)jnum(n.get("left"), 0); nd.right = (int32_t)jnum(n.get("right"), 0);
            nodes.push_back(nd);
        }
    }
    return psk_model_create(ctx, nodes.data(), nodes.size(), first.data(), (uint32_t)first.size(), bias, shrink, feats.data(), (uint32_t)feats.size(), out);
}

psk_status psk_model_load_file(psk_ctx* ctx, const char* path, psk_model** out) {
    if (!ctx || !path || !out) { psk_set_error("model_load: NULL argument"); return PSK_EINVAL; }
    std::ifstream f(path, std::ios::binary);
    if (!f) { psk_set_error("model_load: cannot open %s", path); return PSK_EKEY; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string s = ss.str();
    return psk_model_load_json(ctx, s.data(), s.size(), out);
}

void psk_model_free(psk_model* m) { if (m) { (void)hipSetDevice(m->ctx->device); delete m; } }

psk_status psk_model_info(const psk_model* m, uint32_t* n_trees, uint64_t* n_nodes, uint32_t* n_features) {
    if (!m) { psk_set_error("NULL model"); return PSK_EINVAL; }
    if (n_trees) *n_trees = m->dev.n_trees;
    if (n_nodes) *n_nodes = m->n_nodes;
    if (n_features) *n_features = m->dev.n_features;
    return PSK_OK;
}

psk_status psk_model_predict(const psk_model* m, const float* rows, uint32_t n_rows, float* out) {
    if (!m || (n_rows && (!rows || !out))) { psk_set_error("model_predict: NULL argument"); return PSK_EINVAL; }
    if (!n_rows) return PSK_OK;
    PSK_LANE(lg, m->ctx);
    Lane* ctx = lg.lane;
    const size_t rb = sizeof(float) * (size_t)n_rows * m->dev.n_features, ob = sizeof(float) * (size_t)n_rows;
    PSK_TRY(ctx->q_g.reserve(rb + ob + 256));
    float* d_rows = (float*)ctx->q_g.p; float* d_out = (float*)((char*)ctx->q_g.p + ((rb + 255) & ~(size_t)255));
    PSK_HIP(hipMemcpyAsync(d_rows, rows, rb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(model_predict_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, ctx->stream, m->dev, d_rows, n_rows, d_out);
    PSK_HIP(hipMemcpyAsync(out, d_out, ob, hipMemcpyDeviceToHost, ctx->stream));
    PSK_HIP(hipStreamSynchronize(ctx->stream));
    return PSK_OK;
}

}  // extern "C"
