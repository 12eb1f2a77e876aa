// Multi-GPU exchange behind the C-ABI: packed device-side sketch records, and the two collectives the path has —
// the all-gather of per-shard hit lists (search, all-vs-all) and the all-gather of the shards' sketches (all-vs-all query side) —
// over RCCL on the calling lane's stream. librccl.so is loaded at run time (dlopen): a build or a host without RCCL still gives the
// complete single-GPU library, and these entry points then return PSK_ERCCL ("replicas only", SURVEY.md §8e fallback row).
#include "common.h"
#include <dlfcn.h>

// ------------------------------------------------------------------ device-side sketch records
// A packed sketch is one self-contained byte range of HBM: what an all-gather moves over xGMI as it is, without the
// D2H -> bytes -> H2D hop of psk_sketch_export / psk_sketch_import.
//   [PackHeader 64 B][contig_len u32 x nc][contig_seed_start u32 x (nc+1)] pad16
//   [seed_kmer u32 x ns] pad16 [seed_pos u32 x ns] pad16 [seed_meta u32 x ns] pad16 [markers u64 x nm] pad16
namespace {
struct PackHeader { uint32_t magic, version; int32_t c, marker_c, k; uint32_t has_seeds, n_contigs, reserved; uint64_t n_seeds, n_markers, total_len, bytes; };
static_assert(sizeof(PackHeader) == 64, "PackHeader is 64 bytes");
constexpr uint32_t PACK_MAGIC = 0x4B53504Bu;   // "KPSK"
inline uint64_t al16(uint64_t x) { return (x + 15) & ~15ull; }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct PackLayout { uint64_t o_len, o_cs, o_kmer, o_pos, o_meta, o_mark, end; };
inline PackLayout pack_layout(uint64_t nc, uint64_t ns, uint64_t nm) {
    PackLayout L;
    L.o_len = sizeof(PackHeader); L.o_cs = L.o_len + 4 * nc; L.o_kmer = al16(L.o_cs + 4 * (nc + 1));
    L.o_pos = al16(L.o_kmer + 4 * ns); L.o_meta = al16(L.o_pos + 4 * ns); L.o_mark = al16(L.o_meta + 4 * ns); L.end = al16(L.o_mark + 8 * nm);
    return L;
}
__global__ void build_pm_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ meta, uint64_t* __restrict__ pm, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pm[i] = ((uint64_t)pos[i] << 32) | meta[i];
}

// One launch moves every array of every record of a batch: segment table on the device, workgroup (slice, segment) copies its
// 16 K words of the segment (4-byte granularity: a sketch's slice of a store starts at any seed, i.e. any 4-byte boundary).
struct Seg { const uint32_t* src; uint32_t* dst; uint64_t words; };
constexpr uint32_t SEG_SLICE = 16384;
__global__ __launch_bounds__(256) void seg_copy_kernel(const Seg* __restrict__ segs) {
    const Seg S = segs[blockIdx.y];
    const uint64_t w0 = (uint64_t)blockIdx.x * SEG_SLICE;
    if (w0 >= S.words) return;
    const uint64_t w1 = w0 + SEG_SLICE < S.words ? w0 + SEG_SLICE : S.words;
    for (uint64_t w = w0 + threadIdx.x; w < w1; w += 256) S.dst[w] = S.src[w];
}
psk_status seg_copy(Lane* lane, const std::vector<Seg>& segs, Scratch& table) {
    if (segs.empty()) return PSK_OK;
    uint64_t maxw = 0;
    for (const Seg& s : segs) maxw = std::max(maxw, s.words);
    if (maxw == 0) return PSK_OK;
    PSK_TRY(table.reserve(sizeof(Seg) * segs.size()));
    PSK_HIP(hipMemcpyAsync(table.p, segs.data(), sizeof(Seg) * segs.size(), hipMemcpyHostToDevice, lane->stream));
    for (size_t s0 = 0; s0 < segs.size(); s0 += 65535) {      // grid.y limit
        const uint32_t ny = (uint32_t)std::min<size_t>(65535, segs.size() - s0);
        hipLaunchKernelGGL(seg_copy_kernel, dim3((uint32_t)((maxw + SEG_SLICE - 1) / SEG_SLICE), ny), dim3(256), 0, lane->stream, (const Seg*)table.p + s0);
    }
    return PSK_OK;
}

// n sketches -> n records at d_dst + offsets[i]. ONE header upload (all headers and contig tables, staged contiguously), ONE copy
// launch, ONE synchronisation (the host staging is reused by the next call).
psk_status pack_many_impl(Lane* lane, const psk_sketch* const* sk, uint32_t n, void* d_dst, const uint64_t* offsets, uint64_t capacity) {
    std::vector<PackLayout> L(n);
    uint64_t hdr_bytes = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (!sk[i]) { psk_set_error("pack: NULL sketch %u", i); return PSK_EINVAL; }
        if (offsets[i] & 15) { psk_set_error("pack: record %u is not 16-byte aligned", i); return PSK_EINVAL; }
        L[i] = pack_layout(sk[i]->contig_len.size(), sk[i]->n_seeds, sk[i]->n_markers);
        if (offsets[i] + L[i].end > capacity) { psk_set_error("pack: destination holds %llu bytes, record %u ends at %llu", (unsigned long long)capacity, i, (unsigned long long)(offsets[i] + L[i].end)); return PSK_EINVAL; }
        hdr_bytes += L[i].o_kmer;
    }
    void* hp;
    PSK_TRY(lane->pinned(hdr_bytes + 64, &hp));
    memset(hp, 0, hdr_bytes);
    PSK_TRY(lane->s_misc.reserve(hdr_bytes + 64));
    std::vector<Seg> segs;
    segs.reserve(5 * (size_t)n);
    uint64_t ho = 0;
    char* d = (char*)d_dst;
    for (uint32_t i = 0; i < n; i++) {
        const psk_sketch* s = sk[i];
        const uint64_t nc = s->contig_len.size(), ns = s->n_seeds, nm = s->n_markers;
        PackHeader* H = (PackHeader*)((char*)hp + ho);
        H->magic = PACK_MAGIC; H->version = 1; H->c = s->params.c; H->marker_c = s->params.marker_c; H->k = s->params.k;
        H->has_seeds = s->has_seeds; H->n_contigs = (uint32_t)nc; H->n_seeds = ns; H->n_markers = nm; H->total_len = s->total_len; H->bytes = L[i].end;
        uint32_t* hl = (uint32_t*)((char*)H + L[i].o_len);
        for (uint64_t c = 0; c < nc; c++) hl[c] = s->contig_len[c];
        uint32_t* hc = (uint32_t*)((char*)H + L[i].o_cs);
        for (uint64_t c = 0; c <= nc; c++) hc[c] = c < s->contig_seed_start.size() ? s->contig_seed_start[c] : (uint32_t)ns;
        char* r = d + offsets[i];
        segs.push_back(Seg{(const uint32_t*)((char*)lane->s_misc.p + ho), (uint32_t*)r, L[i].o_kmer / 4});
        if (ns) {
            segs.push_back(Seg{s->store->seed_kmer + s->seed_off, (uint32_t*)(r + L[i].o_kmer), ns});
            segs.push_back(Seg{s->store->seed_pos + s->seed_off, (uint32_t*)(r + L[i].o_pos), ns});
            segs.push_back(Seg{s->store->seed_meta + s->seed_off, (uint32_t*)(r + L[i].o_meta), ns});
        }
        if (nm) segs.push_back(Seg{(const uint32_t*)(s->store->markers + s->marker_off), (uint32_t*)(r + L[i].o_mark), 2 * nm});
        ho += L[i].o_kmer;
    }
    PSK_HIP(hipMemcpyAsync(lane->s_misc.p, hp, hdr_bytes, hipMemcpyHostToDevice, lane->stream));
    PSK_TRY(seg_copy(lane, segs, lane->s_tmp));
    PSK_HIP(hipStreamSynchronize(lane->stream));
    return PSK_OK;
}

// n records at d_src + offsets[i] -> n device-resident sketches sharing one store. Three synchronisations per BATCH (headers,
// contig tables, done), three copy launches; nothing is per record except host bookkeeping.
// capacity: bytes readable at d_src (0: unknown - the caller vouches for the records); sizes: what the sender announced for every record (may be NULL).
// A record header is only believed when it fits both (a truncated or corrupt record from another rank must not steer reads past the buffer).
psk_status unpack_impl(Lane* lane, psk_ctx* ctx, const void* d_src, const uint64_t* offsets, uint32_t n, psk_sketch** out, uint64_t capacity, const uint64_t* sizes) {
    hipStream_t st = lane->stream;
    const char* src = (const char*)d_src;
    for (uint32_t i = 0; i < n; i++) if (offsets[i] & 15) { psk_set_error("unpack: record %u is not 16-byte aligned", i); return PSK_EINVAL; }
    if (capacity) for (uint32_t i = 0; i < n; i++) if (offsets[i] > capacity || capacity - offsets[i] < sizeof(PackHeader)) { psk_set_error("unpack: record %u starts beyond the buffer (%llu bytes)", i, (unsigned long long)capacity); return PSK_EINVAL; }
    // headers: gathered into one staging block, one D2H
    PSK_TRY(lane->s_misc.reserve(sizeof(PackHeader) * (size_t)n));
    std::vector<Seg> segs(n);
    for (uint32_t i = 0; i < n; i++) segs[i] = Seg{(const uint32_t*)(src + offsets[i]), (uint32_t*)((char*)lane->s_misc.p + sizeof(PackHeader) * (size_t)i), sizeof(PackHeader) / 4};
    PSK_TRY(seg_copy(lane, segs, lane->s_tmp));
    std::vector<PackHeader> H(n);
    PSK_HIP(hipMemcpyAsync(H.data(), lane->s_misc.p, sizeof(PackHeader) * (size_t)n, hipMemcpyDeviceToHost, st));
    PSK_HIP(hipStreamSynchronize(st));
    uint64_t tot_c = 0, tot_s = 0, tot_m = 0;
    for (uint32_t i = 0; i < n; i++) {
        const PackHeader& h = H[i];
        if (h.magic != PACK_MAGIC || h.version != 1 || h.k < 1 || h.k > 16 || h.c < 1 || h.marker_c < 1 ||
            h.bytes != pack_layout(h.n_contigs, h.n_seeds, h.n_markers).end) { psk_set_error("unpack: record %u is not a packed sketch", i); return PSK_EINVAL; }
        if (sizes && h.bytes != sizes[i]) { psk_set_error("unpack: record %u says %llu bytes, its sender announced %llu", i, (unsigned long long)h.bytes, (unsigned long long)sizes[i]); return PSK_EINVAL; }
        if (capacity && h.bytes > capacity - offsets[i]) { psk_set_error("unpack: record %u (%llu bytes at %llu) runs past the buffer (%llu bytes)", i, (unsigned long long)h.bytes, (unsigned long long)offsets[i], (unsigned long long)capacity); return PSK_EINVAL; }
        tot_c += h.n_contigs; tot_s += h.n_seeds; tot_m += h.n_markers;
    }
    if (tot_s >= 0x7FFFFFF0ull || tot_m >= 0x7FFFFFF0ull) { psk_set_error("unpack: batch too large for one store; split it"); return PSK_ELIMIT; }
    // contig tables (per record: contig_len[nc], contig_seed_start[nc+1]): one staging block, one D2H
    const size_t meta_words = 2 * (size_t)tot_c + n;
    std::vector<uint32_t> meta(meta_words);
    PSK_TRY(lane->s_misc.reserve(4 * meta_words + 16));
    {
        uint64_t w = 0;
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t nw = 2 * (uint64_t)H[i].n_contigs + 1;
            segs[i] = Seg{(const uint32_t*)(src + offsets[i] + sizeof(PackHeader)), (uint32_t*)lane->s_misc.p + w, nw};
            w += nw;
        }
    }
    PSK_TRY(seg_copy(lane, segs, lane->s_tmp));
    PSK_HIP(hipMemcpyAsync(meta.data(), lane->s_misc.p, 4 * meta_words, hipMemcpyDeviceToHost, st));
    auto store = std::make_shared<SketchStore>();
    store->ctx = ctx;
    const size_t ns = (size_t)tot_s;
    const size_t b_kmer = 0, b_pos = al256(b_kmer + 4 * ns), b_meta = al256(b_pos + 4 * ns), b_pm = al256(b_meta + 4 * ns), b_cs = al256(b_pm + 8 * ns), b_end = al256(b_cs + 4 * (size_t)(tot_c + n));
    PSK_TRY(ctx->pool_alloc(b_end, &store->base, &store->bytes));
    char* sb = (char*)store->base;
    store->seed_kmer = (uint32_t*)(sb + b_kmer); store->seed_pos = (uint32_t*)(sb + b_pos); store->seed_meta = (uint32_t*)(sb + b_meta);
    store->seed_pm = (uint64_t*)(sb + b_pm); store->contig_seed_start = (uint32_t*)(sb + b_cs);
    PSK_TRY(ctx->pool_alloc(8 * ((size_t)tot_m + 1), &store->mbase, &store->mbytes));
    store->markers = (uint64_t*)store->mbase;
    PSK_HIP(hipStreamSynchronize(st));              // meta[] is on the host
    std::vector<uint32_t> cstart(tot_c + n);
    std::vector<std::unique_ptr<psk_sketch>> sk(n);
    segs.clear();
    segs.reserve(4 * (size_t)n);
    uint64_t so = 0, mo = 0, co = 0, w = 0;
    for (uint32_t i = 0; i < n; i++) {
        const PackHeader& h = H[i];
        const PackLayout L = pack_layout(h.n_contigs, h.n_seeds, h.n_markers);
        sk[i].reset(new psk_sketch());
        psk_sketch* s = sk[i].get();
        s->ctx = ctx; s->params = psk_params{h.c, h.marker_c, h.k}; s->has_seeds = h.has_seeds != 0; s->total_len = h.total_len;
        s->store = store; s->seed_off = so; s->n_seeds = h.n_seeds; s->marker_off = mo; s->n_markers = h.n_markers; s->contig_off = co;
        s->contig_len.assign(meta.begin() + w, meta.begin() + w + h.n_contigs);
        s->contig_seed_start.assign(meta.begin() + w + h.n_contigs, meta.begin() + w + 2 * (uint64_t)h.n_contigs + 1);
        for (uint64_t c = 0; c <= h.n_contigs; c++) {
            if (s->contig_seed_start[c] > h.n_seeds || (c && s->contig_seed_start[c] < s->contig_seed_start[c - 1])) { psk_set_error("unpack: record %u has a corrupt contig table", i); return PSK_EINVAL; }
            cstart[co + c] = (uint32_t)(so + s->contig_seed_start[c]);
        }
        const char* r = src + offsets[i];
        if (h.n_seeds) {
            segs.push_back(Seg{(const uint32_t*)(r + L.o_kmer), store->seed_kmer + so, h.n_seeds});
            segs.push_back(Seg{(const uint32_t*)(r + L.o_pos), store->seed_pos + so, h.n_seeds});
            segs.push_back(Seg{(const uint32_t*)(r + L.o_meta), store->seed_meta + so, h.n_seeds});
        }
        if (h.n_markers) segs.push_back(Seg{(const uint32_t*)(r + L.o_mark), (uint32_t*)(store->markers + mo), 2 * h.n_markers});
        so += h.n_seeds; mo += h.n_markers; co += (uint64_t)h.n_contigs + 1; w += 2 * (uint64_t)h.n_contigs + 1;
    }
    PSK_TRY(seg_copy(lane, segs, lane->s_tmp));
    PSK_HIP(hipMemcpyAsync(store->contig_seed_start, cstart.data(), 4 * cstart.size(), hipMemcpyHostToDevice, st));
    if (ns) hipLaunchKernelGGL(build_pm_kernel, dim3((uint32_t)((ns + 255) / 256)), dim3(256), 0, st, store->seed_pos, store->seed_meta, store->seed_pm, (uint64_t)ns);
    PSK_HIP(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < n; i++) out[i] = sk[i].release();
    return PSK_OK;
}

// ------------------------------------------------------------------ RCCL, loaded at run time
struct NcclId { char b[128]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
static_assert(sizeof(NcclId) == PSK_COMM_ID_BYTES, "psk_comm id = ncclUniqueId");
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("PSK_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) { if (!nm || !*nm) continue; R.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (R.h) break; R.why = dlerror() ? dlerror() : ""; }
        if (!R.h) { if (R.why.empty()) R.why = "librccl.so not found"; return; }
        R.GetUniqueId = (int (*)(NcclId*))dlsym(R.h, "ncclGetUniqueId");
        R.CommInitRank = (int (*)(void**, int, NcclId, int))dlsym(R.h, "ncclCommInitRank");
        R.CommDestroy = (int (*)(void*))dlsym(R.h, "ncclCommDestroy");
        R.CommAbort = (int (*)(void*))dlsym(R.h, "ncclCommAbort");
        R.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(R.h, "ncclAllGather");
        R.GetErrorString = (const char* (*)(int))dlsym(R.h, "ncclGetErrorString");
        if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather) { R.why = "librccl.so lacks an expected symbol"; dlclose(R.h); R.h = nullptr; }
    });
    return &R;
}
constexpr int NCCL_UINT8 = 1;      // ncclUint8 (rccl.h:460)
}  // namespace

struct psk_comm {
    psk_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    void* comm = nullptr;
    std::mutex mu;                 // one collective of this communicator at a time
    uint64_t bytes_sent = 0, bytes_received = 0, collectives = 0;
    // control words of the exchange steps (counts, byte totals, status): device and pinned host scratch made with the communicator, so that
    // entering a collective never depends on an allocation that may fail on ONE rank (its peers would wait in the collective for ever)
    uint64_t* d_ctl = nullptr; uint64_t* h_ctl = nullptr;      // 4 words per rank + 4 to send
    bool dead = false;             // a collective failed or was abandoned on this rank: the communicator was aborted, every later call fails at once
};
static void comm_abort(psk_comm* cm) {
    if (cm->dead) return;
    cm->dead = true;
    if (cm->comm && rccl()->CommAbort) { (void)rccl()->CommAbort(cm->comm); cm->comm = nullptr; }
}

#define PSK_NCCL(expr)                                                                                              \
    do {                                                                                                            \
        int _r = (expr);                                                                                            \
        if (_r != 0) { psk_set_error("%s failed: %s", #expr, rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "?"); return PSK_ERCCL; } \
    } while (0)

static psk_status need_rccl() {
    if (!rccl()->h) { psk_set_error("RCCL is not available (%s): multi-GPU is replicas only", rccl()->why.c_str()); return PSK_ERCCL; }
    return PSK_OK;
}

// all-gather of `bytes` bytes per rank, device to device, on the lane's stream
static psk_status all_gather_dev(psk_comm* cm, Lane* lane, const void* send, void* recv, size_t bytes) {
    if (cm->dead) { psk_set_error("the communicator was aborted by an earlier failure"); return PSK_ERCCL; }
    {
        const int r = rccl()->AllGather(send, recv, bytes, NCCL_UINT8, cm->comm, lane->stream);
        if (r != 0) { psk_set_error("ncclAllGather failed: %s (communicator aborted)", rccl()->GetErrorString ? rccl()->GetErrorString(r) : "?"); comm_abort(cm); return PSK_ERCCL; }
    }
    cm->bytes_sent += bytes * (size_t)(cm->world - 1); cm->bytes_received += bytes * (size_t)(cm->world - 1); cm->collectives++;
    return PSK_OK;
}

// Every rank contributes 4 control words (the last one its local status) and reads everybody's: a rank whose local work failed still ENTERS the
// collective, and all ranks leave the exchange step together with the same verdict (a rank that returned on its own left its peers
// waiting in ncclAllGather for ever). words[3] = local status; on return *bad = the first failing rank's status (PSK_OK if none).
static psk_status control_gather(psk_comm* cm, Lane* lane, uint64_t w0, uint64_t w1, uint64_t w2, psk_status local, std::vector<uint64_t>& all, psk_status* bad) {
    const size_t W = (size_t)cm->world;
    hipStream_t st = lane->stream;
    uint64_t* mine = cm->h_ctl + 4 * W;
    mine[0] = w0; mine[1] = w1; mine[2] = w2; mine[3] = (uint64_t)local;
    hipError_t e = hipMemcpyAsync(cm->d_ctl + 4 * W, mine, 32, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { psk_set_error("exchange control words: %s", hipGetErrorString(e)); comm_abort(cm); return PSK_EHIP; }
    PSK_TRY(all_gather_dev(cm, lane, cm->d_ctl + 4 * W, cm->d_ctl, 32));
    e = hipMemcpyAsync(cm->h_ctl, cm->d_ctl, 32 * W, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { psk_set_error("exchange control words: %s", hipGetErrorString(e)); comm_abort(cm); return PSK_EHIP; }
    all.assign(cm->h_ctl, cm->h_ctl + 4 * W);
    *bad = PSK_OK;
    for (size_t r = 0; r < W; r++) if (all[4 * r + 3] != 0) {
        *bad = (psk_status)all[4 * r + 3];
        if ((int)r != cm->rank || local == PSK_OK) psk_set_error("exchange step abandoned: rank %zu reported status %d", r, (int)*bad);      // (the failing rank keeps its own message)
        break;
    }
    return PSK_OK;
}

extern "C" {

psk_status psk_sketch_pack_size(const psk_sketch* s, uint64_t* bytes) {
    if (!s || !bytes) { psk_set_error("pack_size: NULL argument"); return PSK_EINVAL; }
    *bytes = pack_layout(s->contig_len.size(), s->n_seeds, s->n_markers).end;
    return PSK_OK;
}

psk_status psk_sketch_pack_many(const psk_sketch* const* sketches, uint32_t n, void* d_dst, const uint64_t* offsets, uint64_t capacity) {
    if (n && (!sketches || !d_dst || !offsets)) { psk_set_error("pack_many: NULL argument"); return PSK_EINVAL; }
    if (!n) return PSK_OK;
    if (!sketches[0]) { psk_set_error("pack_many: NULL sketch"); return PSK_EINVAL; }
    if (((uintptr_t)d_dst & 15) != 0) { psk_set_error("pack: destination must be 16-byte aligned"); return PSK_EINVAL; }
    PSK_LANE(lg, sketches[0]->ctx);
    return pack_many_impl(lg.lane, sketches, n, d_dst, offsets, capacity);
}

psk_status psk_sketch_pack(const psk_sketch* s, void* d_dst, uint64_t capacity) {
    if (!s || !d_dst) { psk_set_error("pack: NULL argument"); return PSK_EINVAL; }
    const uint64_t zero = 0;
    return psk_sketch_pack_many(&s, 1, d_dst, &zero, capacity);
}

psk_status psk_sketch_unpack(psk_ctx* ctx, const void* d_src, uint64_t capacity, const uint64_t* offsets, uint32_t n, psk_sketch** out) {
    if (!ctx || (n && (!d_src || !offsets || !out))) { psk_set_error("unpack: NULL argument"); return PSK_EINVAL; }
    for (uint32_t i = 0; i < n; i++) out[i] = nullptr;
    if (!n) return PSK_OK;
    PSK_LANE(lg, ctx);
    return unpack_impl(lg.lane, ctx, d_src, offsets, n, out, capacity, nullptr);
}

psk_status psk_comm_unique_id(void* id) {
    if (!id) { psk_set_error("comm_unique_id: NULL argument"); return PSK_EINVAL; }
    PSK_TRY(need_rccl());
    PSK_NCCL(rccl()->GetUniqueId((NcclId*)id));
    return PSK_OK;
}

psk_status psk_comm_create(psk_ctx* ctx, int rank, int world, const void* id, psk_comm** out) {
    if (!ctx || !id || !out) { psk_set_error("comm_create: NULL argument"); return PSK_EINVAL; }
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { psk_set_error("comm_create: rank %d of %d", rank, world); return PSK_EINVAL; }
    PSK_TRY(need_rccl());
    PSK_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<psk_comm> cm(new psk_comm());
    cm->ctx = ctx; cm->rank = rank; cm->world = world;
    NcclId nid;
    memcpy(&nid, id, sizeof nid);
    PSK_HIP(hipMalloc((void**)&cm->d_ctl, 32 * ((size_t)world + 1)));
    { const hipError_t e = hipHostMalloc((void**)&cm->h_ctl, 32 * ((size_t)world + 1), hipHostMallocDefault); if (e != hipSuccess) { (void)hipFree(cm->d_ctl); psk_set_error("hipHostMalloc: %s", hipGetErrorString(e)); return PSK_ENOMEM; } }
    { const int r = rccl()->CommInitRank(&cm->comm, world, nid, rank); if (r != 0) { (void)hipFree(cm->d_ctl); (void)hipHostFree(cm->h_ctl); psk_set_error("ncclCommInitRank failed: %s", rccl()->GetErrorString ? rccl()->GetErrorString(r) : "?"); return PSK_ERCCL; } }
    *out = cm.release();
    return PSK_OK;
}

void psk_comm_destroy(psk_comm* cm) {
    if (!cm) return;
    (void)hipSetDevice(cm->ctx->device);
    if (cm->comm && rccl()->h) (void)rccl()->CommDestroy(cm->comm);
    if (cm->d_ctl) (void)hipFree(cm->d_ctl);
    if (cm->h_ctl) (void)hipHostFree(cm->h_ctl);
    delete cm;
}

psk_status psk_comm_info(const psk_comm* cm, int* rank, int* world, uint64_t* bytes_sent, uint64_t* collectives) {
    if (!cm) { psk_set_error("comm_info: NULL argument"); return PSK_EINVAL; }
    if (rank) *rank = cm->rank;
    if (world) *world = cm->world;
    if (bytes_sent) *bytes_sent = cm->bytes_sent;
    if (collectives) *collectives = cm->collectives;
    return PSK_OK;
}

// All-gather of ragged per-shard hit lists. The records travel as they are: the caller has put the GLOBAL reference index in
// ref_index and the global query index in `reserved`. Two collectives: the counts (8 bytes per rank), then the lists padded to the
// largest count. *all (psk_free) holds the ranks' lists in rank order; counts[r] (world entries, may be NULL) their lengths.
extern "C++" {
template <class H>
static psk_status gather_hits_t(psk_comm* cm, const H* local, uint64_t n_local, H** all, uint64_t* n_all, uint64_t* counts) {
    if (!cm || !all || !n_all) { psk_set_error("gather_hits: NULL argument"); return PSK_EINVAL; }
    *all = nullptr; *n_all = 0;
    std::lock_guard<std::mutex> lk(cm->mu);
    PSK_LANE(lg, cm->ctx);
    Lane* lane = lg.lane;
    hipStream_t st = lane->stream;
    const size_t W = (size_t)cm->world;
    // collective 1: every rank's count - and its verdict on its own arguments (a rank with bad arguments still takes part)
    psk_status mine = PSK_OK;
    if (n_local && !local) { psk_set_error("gather_hits: NULL argument"); mine = PSK_EINVAL; }
    std::vector<uint64_t> ctl; psk_status bad;
    PSK_TRY(control_gather(cm, lane, mine == PSK_OK ? n_local : 0, 0, 0, mine, ctl, &bad));
    if (bad != PSK_OK) return bad;
    std::vector<uint64_t> cnt(W);
    uint64_t maxc = 0, total = 0;
    for (size_t r = 0; r < W; r++) { cnt[r] = ctl[4 * r]; maxc = std::max(maxc, cnt[r]); total += cnt[r]; }
    if (counts) for (size_t r = 0; r < W; r++) counts[r] = cnt[r];
    // local work that can fail (host and device memory, the upload), then ONE more exchange of status words: all ranks enter the payload collective or none
    const size_t row = sizeof(H) * (size_t)maxc;
    H* res = (H*)malloc(sizeof(H) * std::max<uint64_t>(total, 1));
    PoolScratch buf;
    if (!res) { psk_set_error("out of host memory"); mine = PSK_ENOMEM; }
    if (mine == PSK_OK && maxc) mine = buf.reserve(cm->ctx, row * (W + 1) + 256);
    char* d_send = buf.p ? (char*)buf.p + row * W : nullptr;
    if (mine == PSK_OK && maxc && n_local) {
        const hipError_t e = hipMemcpyAsync(d_send, local, sizeof(H) * (size_t)n_local, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { psk_set_error("gather_hits: %s", hipGetErrorString(e)); mine = PSK_EHIP; }
    }
    {
        const psk_status rc = control_gather(cm, lane, 0, 0, 0, mine, ctl, &bad);
        if (rc != PSK_OK || bad != PSK_OK) { free(res); return rc != PSK_OK ? rc : bad; }
    }
    if (maxc) {
        psk_status rc = all_gather_dev(cm, lane, d_send, buf.p, row);
        hipError_t e = hipSuccess;
        uint64_t w = 0;
        for (size_t r = 0; r < W && e == hipSuccess && rc == PSK_OK; r++) {
            if (cnt[r]) e = hipMemcpyAsync(res + w, (char*)buf.p + row * r, sizeof(H) * (size_t)cnt[r], hipMemcpyDeviceToHost, st);
            w += cnt[r];
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st); else (void)hipStreamSynchronize(st);
        if (e != hipSuccess || rc != PSK_OK) { free(res); if (rc == PSK_OK) { psk_set_error("gather_hits: %s", hipGetErrorString(e)); rc = PSK_EHIP; } return rc; }      // (after the last collective: a local matter)
    }
    *all = res; *n_all = total;
    return PSK_OK;
}

}  // extern "C++"
psk_status psk_gather_hits(psk_comm* cm, const psk_hit* local, uint64_t n_local, psk_hit** all, uint64_t* n_all, uint64_t* counts) { return gather_hits_t<psk_hit>(cm, local, n_local, all, n_all, counts); }
psk_status psk_gather_hits_min(psk_comm* cm, const psk_hit_min* local, uint64_t n_local, psk_hit_min** all, uint64_t* n_all, uint64_t* counts) { return gather_hits_t<psk_hit_min>(cm, local, n_local, all, n_all, counts); }

// All-gather of device-resident sketches (the query side of a sharded all-vs-all): every rank contributes n sketches and receives
// everybody's as sketches on ITS GPU. Packed records HBM -> xGMI -> HBM; two collectives per call: (count, bytes) of every rank,
// then one buffer per rank = [u64 sizes[n]] pad16 [records], padded to the widest. *all (psk_free; every entry psk_sketch_free)
// holds the ranks' sketches in rank order, counts[r] their numbers (world entries).
psk_status psk_gather_sketches(psk_comm* cm, const psk_sketch* const* mine, uint32_t n, psk_sketch*** all, uint32_t* counts) {
    if (!cm || !all || !counts) { psk_set_error("gather_sketches: NULL argument"); return PSK_EINVAL; }
    *all = nullptr;
    std::lock_guard<std::mutex> lk(cm->mu);
    PSK_LANE(lg, cm->ctx);
    Lane* lane = lg.lane;
    hipStream_t st = lane->stream;
    const size_t W = (size_t)cm->world;
    // this rank's verdict on its own arguments travels with its (count, bytes): a rank with a NULL sketch still takes part in collective 1
    psk_status local = PSK_OK;
    if (n && !mine) { psk_set_error("gather_sketches: NULL argument"); local = PSK_EINVAL; }
    std::vector<uint64_t> sizes(n), offs(n);
    const uint64_t table = al16(8 * (uint64_t)n);
    uint64_t mybytes = table;
    for (uint32_t i = 0; i < n && local == PSK_OK; i++) {
        if (!mine[i]) { psk_set_error("gather_sketches: NULL sketch %u", i); local = PSK_EINVAL; break; }
        sizes[i] = pack_layout(mine[i]->contig_len.size(), mine[i]->n_seeds, mine[i]->n_markers).end;
        offs[i] = mybytes; mybytes += sizes[i];
    }
    // collective 1: (count, bytes, status) of every rank
    std::vector<uint64_t> ctl; psk_status bad;
    PSK_TRY(control_gather(cm, lane, local == PSK_OK ? n : 0, local == PSK_OK ? mybytes : 16, 0, local, ctl, &bad));
    if (bad != PSK_OK) return bad;
    uint64_t width = 16, n_total = 0;
    for (size_t r = 0; r < W; r++) { counts[r] = (uint32_t)ctl[4 * r]; n_total += ctl[4 * r]; width = std::max(width, al16(ctl[4 * r + 1])); }
    // local work that can fail (the gather buffer, packing), one exchange of status words, then collective 2: the records
    PoolScratch buf;
    local = buf.reserve(cm->ctx, (size_t)width * (W + 1) + 256);
    char* d_send = buf.p ? (char*)buf.p + (size_t)width * W : nullptr;
    if (local == PSK_OK && n) {
        const hipError_t e = hipMemcpyAsync(d_send, sizes.data(), 8 * (size_t)n, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { psk_set_error("gather_sketches: %s", hipGetErrorString(e)); local = PSK_EHIP; }
        else local = pack_many_impl(lane, mine, n, d_send, offs.data(), mybytes);      // synchronises: sizes[] has been read
    }
    {
        const psk_status rc = control_gather(cm, lane, 0, 0, 0, local, ctl, &bad);
        if (rc != PSK_OK) return rc;
        if (bad != PSK_OK) return bad;
    }
    PSK_TRY(all_gather_dev(cm, lane, d_send, buf.p, (size_t)width));
    // size tables of every rank -> record offsets inside the gathered buffer -> one unpack over all of them
    std::vector<uint64_t> all_sizes(n_total ? n_total : 1);
    {
        uint64_t w = 0;
        for (size_t r = 0; r < W; r++) { if (counts[r]) PSK_HIP(hipMemcpyAsync(all_sizes.data() + w, (char*)buf.p + (size_t)width * r, 8 * (size_t)counts[r], hipMemcpyDeviceToHost, st)); w += counts[r]; }
    }
    PSK_HIP(hipStreamSynchronize(st));
    std::vector<uint64_t> roffs(n_total ? n_total : 1);
    {
        uint64_t w = 0;
        for (size_t r = 0; r < W; r++) {
            uint64_t o = (uint64_t)width * r + al16(8 * (uint64_t)counts[r]);
            for (uint32_t j = 0; j < counts[r]; j++) {
                if (all_sizes[w] & 15 || o + all_sizes[w] > (uint64_t)width * (r + 1)) { psk_set_error("gather_sketches: rank %zu sent a corrupt size table", r); return PSK_EINVAL; }
                roffs[w] = o; o += all_sizes[w]; w++;
            }
        }
    }
    psk_sketch** res = (psk_sketch**)malloc(sizeof(psk_sketch*) * std::max<uint64_t>(n_total, 1));
    if (!res) { psk_set_error("out of host memory"); return PSK_ENOMEM; }
    psk_status rc = PSK_OK;
    // one store per 2^30 seeds' worth of records (unpack's 32-bit offsets)
    for (uint64_t b = 0; b < n_total && rc == PSK_OK;) {
        uint64_t e = b, bytes = 0;
        while (e < n_total && (e == b || bytes + all_sizes[e] < (8ull << 30))) { bytes += all_sizes[e]; e++; }
        rc = unpack_impl(lane, cm->ctx, buf.p, roffs.data() + b, (uint32_t)(e - b), res + b, (uint64_t)width * W, all_sizes.data() + b);
        if (rc != PSK_OK) for (uint64_t i = 0; i < b; i++) { delete res[i]; }
        b = e;
    }
    if (rc != PSK_OK) { free(res); return rc; }
    *all = res;
    return PSK_OK;
}

}  // extern "C"
