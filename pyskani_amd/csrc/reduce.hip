// Chain stage 4: per-pair ANI (mean / median / trimmed mean of the chunk identities), aligned fractions, the psk_hit record.
#include "chain_stages.h"
#include <cmath>

// ------------------------------------------------------------------ per-pair ANI / AF

// pairs without a chunk table (fewer than MIN_ANCHORS anchors: every rescued short contig against an unrelated reference): one
// empty record each, one lane per pair
__global__ __launch_bounds__(256) void pair_empty_kernel(ReduceArgs R, uint32_t n_pairs) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs || R.n_chunks[p] != 0) return;
    psk_hit h{};
    h.ani = -1.0f; h.ani_raw = -1.0f;
    h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
    h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
    R.hits[p] = h;
}

// Two instantiations, like the selection: CAP = RED_SMALL for the bulk (12 KB of LDS instead of 49: residency: 4.1 -> 3.4 ms per 10^5
// pairs of 5 Mb genomes), CAP = RED_CAP for the pairs with more chunk rows than that (and, beyond RED_CAP values, the global sort).
template <int CAP>
__device__ void pair_reduce_pair(const ReduceArgs& R, const uint32_t p) {
    __shared__ double s_v[CAP];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_acc[5];
    const uint32_t nc = R.n_chunks[p];
    if (CAP == RED_SMALL ? nc > (uint32_t)RED_SMALL : nc <= (uint32_t)RED_SMALL) return;      // the other instantiation's pair
    if (R.small_done && nc != 0 && nc <= 64) return;      // pair_reduce_small_kernel took it
    if (R.wave_done && nc > 64 && nc <= 64u * RW_PER) return;      // pair_reduce_wave_kernel took it
    if (nc == 0) {      // only reached when the launch visits every pair (no live list): the empty record of pair_empty_kernel
        if (threadIdx.x == 0) {
            psk_hit h{};
            h.ani = -1.0f; h.ani_raw = -1.0f;
            h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
            h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
            R.hits[p] = h;
        }
        return;
    }
    const ChunkOut* co = R.chunks + (size_t)R.cbase[p];
    if (threadIdx.x == 0) { s_n = 0; for (int i = 0; i < 5; i++) s_acc[i] = 0; }
    __syncthreads();
    // integer totals (order-free) and the number of chunks that kept a chain
    unsigned long long t_cq = 0, t_cr = 0, t_a = 0, t_s = 0, t_i = 0; uint32_t t_m = 0;
    for (uint32_t i = threadIdx.x; i < nc; i += blockDim.x) { const uint32_t ni = co[i].n_intervals; t_cq += co[i].cov_q; t_cr += co[i].cov_q; t_a += co[i].anchors; t_s += ni ? co[i].seeds : 0; t_i += ni; t_m += ni != 0; }
    atomicAdd(&s_acc[0], t_cq); atomicAdd(&s_acc[1], t_cr); atomicAdd(&s_acc[2], t_a); atomicAdd(&s_acc[3], t_s); atomicAdd(&s_acc[4], t_i);
    atomicAdd(&s_n, t_m);
    __syncthreads();
    // their rows compacted in chunk order (the oracle's summation order), 256 rows per step: ballot ranks inside a wave, the four
    // wave totals through LDS. Not needed beyond RED_CAP values (those pairs never index s_idx).
    __shared__ uint32_t s_idx[CAP];
    __shared__ uint32_t s_wt[2][4];
    if (s_n <= (uint32_t)CAP) {
        uint32_t run = 0;
        const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (uint32_t i0 = 0, it = 0; i0 < nc; i0 += 256, it++) {
            const uint32_t i = i0 + threadIdx.x;
            const bool f = i < nc && co[i].n_intervals != 0;
            const unsigned long long bal = __ballot(f);
            if (lane == 0) s_wt[it & 1][w] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, tot = 0;
            for (uint32_t x = 0; x < 4; x++) { const uint32_t c = s_wt[it & 1][x]; if (x < w) before += c; tot += c; }
            if (f) s_idx[run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = i;
            run += tot;
        }
    }
    __syncthreads();
    const uint32_t m = s_n;
    psk_hit h{};
    h.ani = -1.0f;
    const bool overflow = m > (uint32_t)CAP;
    double mean_serial = 0;   // only thread 0 uses it
    if (!overflow) {
        for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) {
            const ChunkOut c = co[s_idx[j]];
            double ratio = (double)c.anchors / (double)(c.seeds > 1 ? c.seeds - 1 : 1);   // end seeds are anchors by construction
            if (ratio > 1.0) ratio = 1.0;
            s_v[j] = pow(ratio, 1.0 / (double)R.k);
        }
        __syncthreads();
        if (R.median || R.robust) {   // bitonic sort of s_v[0..m) padded with +inf
            uint32_t P = 1; while (P < m) P <<= 1;
            for (uint32_t j = m + threadIdx.x; j < P; j += blockDim.x) s_v[j] = INFINITY;
            __syncthreads();
            for (uint32_t kk = 2; kk <= P; kk <<= 1)
                for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
                    for (uint32_t t = threadIdx.x; t < P; t += blockDim.x) {
                        uint32_t ixj = t ^ jj;
                        if (ixj > t) {
                            double a = s_v[t], b = s_v[ixj];
                            bool up = (t & kk) == 0;
                            if ((a > b) == up) { s_v[t] = b; s_v[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
        }
    } else if (R.median || R.robust) {   // very long genomes: sort the chunk values in global scratch
        double* gv = R.big_vals + 2 * (size_t)R.cbase[p] + 1024 * (size_t)p;
        uint32_t P = 1024; while (P < nc) P <<= 1;
        for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
            double v = INFINITY;
            if (i < nc && co[i].n_intervals) {
                double ratio = (double)co[i].anchors / (double)(co[i].seeds > 1 ? co[i].seeds - 1 : 1); if (ratio > 1.0) ratio = 1.0;
                v = pow(ratio, 1.0 / (double)R.k);
            }
            gv[i] = v;
        }
        __syncthreads();
        for (uint32_t kk = 2; kk <= P; kk <<= 1)
            for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
                for (uint32_t t = threadIdx.x; t < P; t += blockDim.x) {
                    uint32_t ixj = t ^ jj;
                    if (ixj > t) {
                        double a = gv[t], b = gv[ixj];
                        bool up = (t & kk) == 0;
                        if ((a > b) == up) { gv[t] = b; gv[ixj] = a; }
                    }
                }
                __syncthreads();
            }
    }
    __syncthreads();
    // mean and sample standard deviation of ALL chunk values (feature of the learned-ANI regression; also the mean of
    // pairs beyond RED_CAP chunks): two block-parallel passes, fixed thread -> element mapping (deterministic)
    __shared__ double s_red[8];
    auto block_sum = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        return s_red[0] + s_red[1] + s_red[2] + s_red[3];
    };
    auto chunk_val = [&](uint32_t i) {
        double ratio = (double)co[i].anchors / (double)(co[i].seeds > 1 ? co[i].seeds - 1 : 1); if (ratio > 1.0) ratio = 1.0;
        return pow(ratio, 1.0 / (double)R.k);
    };
    double part = 0;
    if (!overflow) { for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) part += s_v[j]; }
    else { for (uint32_t i = threadIdx.x; i < nc; i += blockDim.x) if (co[i].n_intervals) part += chunk_val(i); }
    const double mean_all = m ? block_sum(part) / (double)m : 0.0;
    part = 0;
    if (!overflow) { for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) { const double d = s_v[j] - mean_all; part += d * d; } }
    else { for (uint32_t i = threadIdx.x; i < nc; i += blockDim.x) if (co[i].n_intervals) { const double d = chunk_val(i) - mean_all; part += d * d; } }
    const double ssq = block_sum(part);
    const double std_all = m > 1 ? sqrt(ssq / (double)(m - 1)) : 0.0;
    mean_serial = mean_all;
    if (threadIdx.x == 0) {
        h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
        h.n_chunks = m; h.n_intervals = (uint32_t)s_acc[4];
        h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
        h.covered_query = s_acc[0]; h.covered_ref = s_acc[1]; h.sum_chain_anchors = s_acc[2]; h.sum_chunk_seeds = s_acc[3];
        if (m > 0) {
            double ani;
            bool ok = true;
            if (overflow && (R.median || R.robust)) {
                const double* gv = R.big_vals + 2 * (size_t)R.cbase[p] + 1024 * (size_t)p;
                if (R.median) ani = gv[m / 2];
                else {
                    uint32_t lo = 0, hi = m;
                    if (m - 2 * (m / 10) > 0) { lo = m / 10; hi = m - m / 10; }
                    double sum = 0; for (uint32_t i = lo; i < hi; i++) sum += gv[i];
                    ani = sum / (double)(hi - lo);
                }
            }
            else if (overflow) ani = mean_serial;
            else if (R.median) ani = s_v[m / 2];
            else {
                uint32_t lo = 0, hi = m;
                if (R.robust && m - 2 * (m / 10) > 0) { lo = m / 10; hi = m - m / 10; }
                double sum = 0; for (uint32_t i = lo; i < hi; i++) sum += s_v[i];
                ani = sum / (double)(hi - lo);
            }
            double afq = (double)s_acc[0] / (double)R.pairs[p].q_total_len; if (afq > 1) afq = 1;
            double afr = (double)s_acc[0] / (double)R.pairs[p].r_total_len; if (afr > 1) afr = 1;   // one covered-bases count serves both
            h.af_query = (float)afq; h.af_ref = (float)afr;
            if (ok && (afq >= R.min_af || afr >= R.min_af)) h.ani = (float)ani;
            h.ani_raw = h.ani; h.ani_std = (float)std_all;
        }
        R.hits[p] = h;
    }
}
// Pairs whose chunk table has at most 64 rows (short contigs: 1-3 chunks) - ONE WAVE per pair, a lane per chunk, shuffles instead
// of LDS and workgroup barriers; four independent pairs per workgroup. Same arithmetic and summation order as pair_reduce_pair
// (values in chunk order for the mean, ascending for median / trimmed mean).
__global__ __launch_bounds__(256) void pair_reduce_small_kernel(ReduceArgs R, uint32_t n_pairs) {
    __shared__ double s_sorted[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = R.live ? *R.n_live : n_pairs;
    for (uint32_t k = blockIdx.x * 4 + wave; k < n; k += gridDim.x * 4) {
        const uint32_t p = R.live ? R.live[k] : k;
        const uint32_t nc = R.n_chunks[p];
        if (nc == 0 && !R.live) {                         // a launch without the live list (few pairs): the empty record here, as pair_empty_kernel writes it
            if (lane == 0) {
                psk_hit h{};
                h.ani = -1.0f; h.ani_raw = -1.0f;
                h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
                h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
                R.hits[p] = h;
            }
            continue;
        }
        if (nc == 0 || nc > 64) continue;                 // empty records / larger tables: the other kernels
        if (R.tiny_done && nc <= 4) continue;             // pair_reduce_tiny_kernel took it
        const ChunkOut* co = R.chunks + (size_t)R.cbase[p];
        ChunkOut c{};
        if ((uint32_t)lane < nc) c = co[lane];
        const bool valid = (uint32_t)lane < nc && c.n_intervals != 0;
        unsigned long long t_cq = c.cov_q, t_a = c.anchors, t_s = valid ? c.seeds : 0, t_i = c.n_intervals;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t_cq += __shfl_xor(t_cq, o); t_a += __shfl_xor(t_a, o); t_s += __shfl_xor(t_s, o); t_i += __shfl_xor(t_i, o); }
        const unsigned long long vm = __ballot(valid);
        const uint32_t m = (uint32_t)__popcll(vm);
        double v = 0.0;
        if (valid) {
            double ratio = (double)c.anchors / (double)(c.seeds > 1 ? c.seeds - 1 : 1);   // end seeds are anchors by construction
            if (ratio > 1.0) ratio = 1.0;
            v = pow(ratio, 1.0 / (double)R.k);
        }
        // compact the values in chunk order (position = number of valid lanes below)
        const uint32_t pos = (uint32_t)__popcll(vm & ((1ull << lane) - 1));
        double* sv = s_sorted[wave];
        lds_wave_sync();
        if (valid) sv[pos] = v;
        lds_wave_sync();
        // mean and sample standard deviation of all values
        double sum_all = 0;
        for (uint32_t j = 0; j < m; j++) sum_all += sv[j];                 // chunk order, like the serial sum of the big path
        const double mean_all = m ? sum_all / (double)m : 0.0;
        double dev = valid ? (v - mean_all) * (v - mean_all) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dev += __shfl_xor(dev, o);
        const double std_all = m > 1 ? sqrt(dev / (double)(m - 1)) : 0.0;
        double ani = mean_all;
        if ((R.median || R.robust) && m) {
            // ascending order: rank = values below + equal values at lower positions
            uint32_t rank = 0;
            const double mine = (uint32_t)lane < m ? sv[lane] : 0.0;
            for (uint32_t j = 0; j < m; j++) { const double o = sv[j]; rank += (o < mine) || (o == mine && j < (uint32_t)lane); }
            lds_wave_sync();
            if ((uint32_t)lane < m) sv[rank] = mine;
            lds_wave_sync();
            if (R.median) ani = sv[m / 2];
            else {
                uint32_t lo = 0, hi = m;
                if (m - 2 * (m / 10) > 0) { lo = m / 10; hi = m - m / 10; }
                double sum = 0; for (uint32_t j = lo; j < hi; j++) sum += sv[j];
                ani = sum / (double)(hi - lo);
            }
        }
        if (lane == 0) {
            psk_hit h{};
            h.ani = -1.0f;
            h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
            h.n_chunks = m; h.n_intervals = (uint32_t)t_i;
            h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
            h.covered_query = t_cq; h.covered_ref = t_cq; h.sum_chain_anchors = t_a; h.sum_chunk_seeds = t_s;
            if (m > 0) {
                double afq = (double)t_cq / (double)R.pairs[p].q_total_len; if (afq > 1) afq = 1;
                double afr = (double)t_cq / (double)R.pairs[p].r_total_len; if (afr > 1) afr = 1;   // one covered-bases count serves both
                h.af_query = (float)afq; h.af_ref = (float)afr;
                if (afq >= R.min_af || afr >= R.min_af) h.ani = (float)ani;
                h.ani_raw = h.ani; h.ani_std = (float)std_all;
            }
            R.hits[p] = h;
        }
    }
}

// Contig pairs: one to three chunks. A wave per pair leaves 61 lanes idle for 17 M pairs per metagenome step (15 ms); here ONE LANE reduces a pair of up to four chunk
// rows (mean ANI only: median / trimmed mean stay with the wave kernel). Same values in the same order as pair_reduce_small_kernel: the mean as the sequential sum
// in chunk order, the squared deviations added the way that kernel's shuffle tree adds lanes 0..3: (d0 + d2) + (d1 + d3).
__global__ __launch_bounds__(256) void pair_reduce_tiny_kernel(ReduceArgs R, uint32_t n_pairs) {
    const uint32_t n = R.live ? *R.n_live : n_pairs;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = R.live ? R.live[k] : k;
    const uint32_t nc = R.n_chunks[p];
    if (nc == 0 || nc > 4) return;
    const ChunkOut* co = R.chunks + (size_t)R.cbase[p];
    unsigned long long t_cq = 0, t_a = 0, t_s = 0, t_i = 0;
    double v[4] = {0.0, 0.0, 0.0, 0.0}; bool valid[4] = {false, false, false, false};
    uint32_t m = 0;
    double sum_all = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) if (r < nc) {
        const ChunkOut c = co[r];
        valid[r] = c.n_intervals != 0;
        t_cq += c.cov_q; t_a += c.anchors; t_s += valid[r] ? c.seeds : 0; t_i += c.n_intervals;
        if (valid[r]) {
            double ratio = (double)c.anchors / (double)(c.seeds > 1 ? c.seeds - 1 : 1);   // end seeds are anchors by construction
            if (ratio > 1.0) ratio = 1.0;
            v[r] = pow(ratio, 1.0 / (double)R.k);
            sum_all += v[r];
            m++;
        }
    }
    const double mean_all = m ? sum_all / (double)m : 0.0;
    double d[4];
#pragma unroll
    for (int r = 0; r < 4; r++) d[r] = valid[r] ? (v[r] - mean_all) * (v[r] - mean_all) : 0.0;
    const double dev = (d[0] + d[2]) + (d[1] + d[3]);
    const double std_all = m > 1 ? sqrt(dev / (double)(m - 1)) : 0.0;
    psk_hit h{};
    h.ani = -1.0f;
    h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
    h.n_chunks = m; h.n_intervals = (uint32_t)t_i;
    h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
    h.covered_query = t_cq; h.covered_ref = t_cq; h.sum_chain_anchors = t_a; h.sum_chunk_seeds = t_s;
    if (m > 0) {
        double afq = (double)t_cq / (double)R.pairs[p].q_total_len; if (afq > 1) afq = 1;
        double afr = (double)t_cq / (double)R.pairs[p].r_total_len; if (afr > 1) afr = 1;   // one covered-bases count serves both
        h.af_query = (float)afq; h.af_ref = (float)afr;
        if (afq >= R.min_af || afr >= R.min_af) h.ani = (float)mean_all;
        h.ani_raw = h.ani; h.ani_std = (float)std_all;
    }
    R.hits[p] = h;
}

// Pairs whose chunk table has 65 .. 64 RW_PER rows (a pair of 5 Mb genomes: ~170-300 chunks): ONE WAVE per pair, RW_PER rows per lane, four independent pairs per
// workgroup, no workgroup barrier - the workgroup-per-pair kernel spends its time in a dozen barriers and a one-thread sum over LDS while 255 threads
// wait (33 ns per pair of a 10^6-pair batch). Same arithmetic in the same order as pair_reduce_pair: the chunk values compacted in chunk order, mean and
// deviation sums as that kernel's 256 threads form them (one value per thread, a shuffle tree per 64, the four trees added in order), the ANI mean as the
// sequential sum in chunk order - here over lane reads of registers -, median / trimmed mean over an ascending order.
__global__ __launch_bounds__(256) void pair_reduce_wave_kernel(ReduceArgs R, uint32_t n_pairs) {
    __shared__ double s_val[4][64 * RW_PER];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = R.live ? *R.n_live : n_pairs;
    double* sv = s_val[wave];
    for (uint32_t k = blockIdx.x * 4 + wave; k < n; k += gridDim.x * 4) {
        const uint32_t p = R.live ? R.live[k] : k;
        const uint32_t nc = R.n_chunks[p];
        if (nc <= 64 || nc > 64u * RW_PER) continue;      // the one-wave-one-row kernel / the workgroup kernel
        const ChunkOut* co = R.chunks + (size_t)R.cbase[p];
        unsigned long long t_cq = 0, t_a = 0, t_s = 0, t_i = 0;
        double v[RW_PER]; bool valid[RW_PER];
#pragma unroll
        for (int g = 0; g < RW_PER; g++) {
            const uint32_t r = 64u * g + (uint32_t)lane;
            ChunkOut c{};
            if (r < nc) c = co[r];
            valid[g] = r < nc && c.n_intervals != 0;
            t_cq += c.cov_q; t_a += c.anchors; t_s += valid[g] ? c.seeds : 0; t_i += c.n_intervals;
            v[g] = 0.0;
            if (valid[g]) {
                double ratio = (double)c.anchors / (double)(c.seeds > 1 ? c.seeds - 1 : 1);   // end seeds are anchors by construction
                if (ratio > 1.0) ratio = 1.0;
                v[g] = pow(ratio, 1.0 / (double)R.k);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t_cq += __shfl_xor(t_cq, o); t_a += __shfl_xor(t_a, o); t_s += __shfl_xor(t_s, o); t_i += __shfl_xor(t_i, o); }
        // the values of the chunks that kept a chain, compacted in chunk order
        lds_wave_sync();
        uint32_t m = 0;
#pragma unroll
        for (int g = 0; g < RW_PER; g++) {
            const unsigned long long vm = __ballot(valid[g]);
            if (valid[g]) sv[m + (uint32_t)__popcll(vm & ((1ull << lane) - 1))] = v[g];
            m += (uint32_t)__popcll(vm);
        }
        lds_wave_sync();
        double cv[RW_PER];      // compacted value j sits where thread j of the workgroup kernel has it: lane j & 63 of group j >> 6
#pragma unroll
        for (int g = 0; g < RW_PER; g++) cv[g] = 64u * g + (uint32_t)lane < m ? sv[64 * g + lane] : 0.0;
        auto tree4 = [&](const double* x) {      // block_sum of pair_reduce_pair: a shuffle tree per 64 threads, the four results added in order
            // (thread t of that kernel's 256 adds elements t and t + 256 before the tree: groups g and g + 4 here)
            double t[4];
#pragma unroll
            for (int g = 0; g < 4; g++) { double y = 0.0 + x[g]; y += x[g + 4]; for (int o = 32; o > 0; o >>= 1) y += __shfl_xor(y, o); t[g] = y; }
            return t[0] + t[1] + t[2] + t[3];
        };
        const double mean_all = m ? tree4(cv) / (double)m : 0.0;
        double dv[RW_PER];
#pragma unroll
        for (int g = 0; g < RW_PER; g++) { const double d = cv[g] - mean_all; dv[g] = 64u * g + (uint32_t)lane < m ? 0.0 + d * d : 0.0; }
        const double ssq = tree4(dv);
        const double std_all = m > 1 ? sqrt(ssq / (double)(m - 1)) : 0.0;
        double ani = 0.0;
        if (m) {
            if (R.median || R.robust) {
                // ascending order: rank = values below + equal values at lower positions
                uint32_t rank[RW_PER];
#pragma unroll
                for (int g = 0; g < RW_PER; g++) rank[g] = 0;
                for (uint32_t j = 0; j < m; j++) {
                    const double o = sv[j];
#pragma unroll
                    for (int g = 0; g < RW_PER; g++) rank[g] += (o < cv[g]) || (o == cv[g] && j < 64u * g + (uint32_t)lane);
                }
                lds_wave_sync();
#pragma unroll
                for (int g = 0; g < RW_PER; g++) if (64u * g + (uint32_t)lane < m) sv[rank[g]] = cv[g];
                lds_wave_sync();
                if (R.median) ani = sv[m / 2];
                else {
                    uint32_t lo = 0, hi = m;
                    if (m - 2 * (m / 10) > 0) { lo = m / 10; hi = m - m / 10; }
                    double sum = 0; for (uint32_t j = lo; j < hi; j++) sum += sv[j];
                    ani = sum / (double)(hi - lo);
                }
            } else {
                // the sequential sum in chunk order, over lane reads (a row without a chain contributes an exact + 0.0)
                double sum = 0;
#pragma unroll
                for (int g = 0; g < RW_PER; g++) {
                    const uint32_t lo32 = (uint32_t)__double2loint(v[g]), hi32 = (uint32_t)__double2hiint(v[g]);
                    const uint32_t cnt = nc > 64u * g ? (nc - 64u * g < 64u ? nc - 64u * g : 64u) : 0u;
                    for (uint32_t l = 0; l < cnt; l++)
                        sum += __hiloint2double((int)__builtin_amdgcn_readlane(hi32, l), (int)__builtin_amdgcn_readlane(lo32, l));
                }
                ani = sum / (double)m;
            }
        }
        if (lane == 0) {
            psk_hit h{};
            h.ani = -1.0f;
            h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
            h.n_chunks = m; h.n_intervals = (uint32_t)t_i;
            h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
            h.covered_query = t_cq; h.covered_ref = t_cq; h.sum_chain_anchors = t_a; h.sum_chunk_seeds = t_s;
            if (m > 0) {
                double afq = (double)t_cq / (double)R.pairs[p].q_total_len; if (afq > 1) afq = 1;
                double afr = (double)t_cq / (double)R.pairs[p].r_total_len; if (afr > 1) afr = 1;   // one covered-bases count serves both
                h.af_query = (float)afq; h.af_ref = (float)afr;
                if (afq >= R.min_af || afr >= R.min_af) h.ani = (float)ani;
                h.ani_raw = h.ani; h.ani_std = (float)std_all;
            }
            R.hits[p] = h;
        }
    }
}

__global__ __launch_bounds__(256) void pair_reduce_kernel(ReduceArgs R, uint32_t n_pairs) {      // one workgroup per LIVE pair, fixed grid over the list
    const uint32_t n = R.live ? *R.n_live : n_pairs;        // small launches skip the list: every pair is visited
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        pair_reduce_pair<RED_SMALL>(R, R.live ? R.live[k] : k);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void pair_reduce_large_kernel(ReduceArgs R, uint32_t n_pairs) {      // the pairs with more than RED_SMALL chunk rows
    const uint32_t n = R.live ? *R.n_live : n_pairs;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        pair_reduce_pair<RED_CAP>(R, R.live ? R.live[k] : k);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void pair_ref_keys_kernel(const uint2* __restrict__ pair_qr, uint32_t n_pairs, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pairs) { keys[p] = pair_qr[p].y; ids[p] = p; }
}
