// rust/ffi.rs — the `extern "C"` binding a pyskani maintainer adds as src/pyskani/_skani/ffi.rs to put
// libpyskani_amd.so behind the existing PyO3 classes (INTEGRATION.md section 2 lists the four call sites to swap in lib.rs).
// NOT compiled in this repository: the build image has no Rust toolchain. tests/test_api_cpu.py
// checks that this file declares exactly the symbols of include/pyskani_amd.h.
// Every symbol of include/pyskani_amd.h (tests/test_api_cpu.py holds the header, this block's Python twin
// pyskani_amd/_capi.py and the built library to the same symbol set).
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct PskCtx { _p: [u8; 0] }
#[repr(C)] pub struct PskSketch { _p: [u8; 0] }
#[repr(C)] pub struct PskDb { _p: [u8; 0] }
#[repr(C)] pub struct PskModel { _p: [u8; 0] }
#[repr(C)] pub struct PskComm { _p: [u8; 0] }
pub const PSK_COMM_ID_BYTES: usize = 128;

#[repr(C)] #[derive(Clone, Copy)]
pub struct PskParams { pub c: i32, pub marker_c: i32, pub k: i32 }

#[repr(C)] #[derive(Clone, Copy)]
pub struct PskQueryOpts {
    pub learned_ani: i32, pub median: i32, pub robust: i32, pub faster_small: i32,
    pub cutoff: f64, pub min_aligned_frac: f64,
    pub model: *const PskModel,          // `model_opt` of lib.rs:614 / :650; null = none
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct PskHit {
    pub ani: f32, pub af_query: f32, pub af_ref: f32, pub ref_index: u32,
    pub n_chunks: u32, pub n_intervals: u32,
    pub n_anchors: u64, pub covered_query: u64, pub covered_ref: u64,
    pub sum_chain_anchors: u64, pub sum_chunk_seeds: u64,
    pub ani_raw: f32, pub ani_std: f32, pub learned: u32, pub reserved: u32,
}
/// psk_hit_min: what `Hit` holds (hit.rs:77-104) in 20 bytes; `query` = index of the query within the call, bit 31 = the model produced `ani`
#[repr(C)] #[derive(Clone, Copy)]
pub struct PskHitMin { pub ani: f32, pub af_query: f32, pub af_ref: f32, pub ref_index: u32, pub query: u32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct PskSeed { pub kmer: u32, pub pos: u32, pub contig: u32, pub canon: u32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct PskTreeNode { pub feature: i32, pub threshold: f32, pub left: i32, pub right: i32,
                         pub value: f32, pub missing: i32, pub is_leaf: i32, pub reserved: i32 }

#[link(name = "pyskani_amd")]
extern "C" {
    // errors, version, memory
    pub fn psk_last_error() -> *const c_char;
    pub fn psk_version() -> *const c_char;
    pub fn psk_abi_version() -> c_int;      // compare with PSK_ABI_VERSION (5) before anything else is called
    pub fn psk_free(p: *mut c_void);
    // context (one per GPU)
    pub fn psk_ctx_create(device: c_int, out: *mut *mut PskCtx) -> c_int;
    pub fn psk_ctx_destroy(ctx: *mut PskCtx);
    pub fn psk_ctx_synchronize(ctx: *mut PskCtx) -> c_int;
    pub fn psk_ctx_set_timing(ctx: *mut PskCtx, on: c_int) -> c_int;
    pub fn psk_ctx_timing(ctx: *mut PskCtx, kernel: *const c_char, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn psk_ctx_clock_probe(ctx: *mut PskCtx, mhz: *mut f64, ms: *mut f64) -> c_int;
    pub fn psk_ctx_work(ctx: *mut PskCtx, pairs: *mut u64, items: *mut u64, anchors: *mut u64, reset: c_int) -> c_int;
    pub fn psk_ctx_join_work(ctx: *mut PskCtx, lookups: *mut u64, visited: *mut u64, candidates: *mut u64, rows: *mut u64, reset: c_int) -> c_int;
    pub fn psk_device_alloc(ctx: *mut PskCtx, bytes: usize, dptr: *mut *mut c_void) -> c_int;
    pub fn psk_device_free(ctx: *mut PskCtx, dptr: *mut c_void) -> c_int;
    pub fn psk_memcpy_h2d(ctx: *mut PskCtx, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    // sketching: Database::_sketch (lib.rs:140-185)
    pub fn psk_sketch_host(ctx: *mut PskCtx, p: *const PskParams, contigs: *const *const u8,
                           lens: *const u64, n_contigs: u32, want_seeds: c_int,
                           out: *mut *mut PskSketch) -> c_int;
    pub fn psk_sketch_many_host(ctx: *mut PskCtx, p: *const PskParams, contigs: *const *const u8, lens: *const u64,
                                genome_first_contig: *const u32, n_genomes: u32, want_seeds: c_int,
                                out: *mut *mut PskSketch) -> c_int;
    pub fn psk_sketch_batch_device(ctx: *mut PskCtx, p: *const PskParams, d_bases: *const u8, contig_off: *const u64,
                                   contig_len: *const u64, genome_first_contig: *const u32, n_genomes: u32,
                                   want_seeds: c_int, out: *mut *mut PskSketch) -> c_int;
    pub fn psk_sketch_free(s: *mut PskSketch);
    pub fn psk_sketch_free_many(sketches: *const *mut PskSketch, n: u32);
    // (de)serialisation hooks for `Database.save/load/open` (lib.rs:249-337): plain arrays in, plain arrays out
    pub fn psk_sketch_info(s: *const PskSketch, p: *mut PskParams, n_seeds: *mut u64, n_markers: *mut u64,
                           total_len: *mut u64, n_contigs: *mut u32) -> c_int;
    pub fn psk_sketch_export(s: *const PskSketch, seeds: *mut PskSeed, markers: *mut u64) -> c_int;
    pub fn psk_sketch_contig_lens(s: *const PskSketch, lens: *mut u32) -> c_int;
    pub fn psk_sketch_import(ctx: *mut PskCtx, p: *const PskParams, contig_lens: *const u32, n_contigs: u32,
                             seeds: *const PskSeed, n_seeds: u64, markers: *const u64, n_markers: u64,
                             has_seeds: c_int, out: *mut *mut PskSketch) -> c_int;
    // device-side records for the multi-GPU exchange (an RCCL all-gather moves them as they are)
    pub fn psk_sketch_pack_size(s: *const PskSketch, bytes: *mut u64) -> c_int;
    pub fn psk_sketch_pack(s: *const PskSketch, d_dst: *mut c_void, capacity: u64) -> c_int;
    pub fn psk_sketch_unpack(ctx: *mut PskCtx, d_src: *const c_void, capacity: u64, offsets: *const u64, n: u32,
                             out: *mut *mut PskSketch) -> c_int;
    pub fn psk_sketch_pack_many(sketches: *const *const PskSketch, n: u32, d_dst: *mut c_void, offsets: *const u64, capacity: u64) -> c_int;
    // multi-GPU exchange over RCCL (one process per GPU; the id travels over whatever channel the host program has)
    pub fn psk_comm_unique_id(id: *mut c_void) -> c_int;
    pub fn psk_comm_create(ctx: *mut PskCtx, rank: c_int, world: c_int, id: *const c_void, out: *mut *mut PskComm) -> c_int;
    pub fn psk_comm_destroy(comm: *mut PskComm);
    pub fn psk_comm_info(comm: *const PskComm, rank: *mut c_int, world: *mut c_int, bytes_sent: *mut u64, collectives: *mut u64) -> c_int;
    pub fn psk_gather_hits(comm: *mut PskComm, local: *const PskHit, n_local: u64, all: *mut *mut PskHit, n_all: *mut u64, counts: *mut u64) -> c_int;
    pub fn psk_gather_hits_min(comm: *mut PskComm, local: *const PskHitMin, n_local: u64, all: *mut *mut PskHitMin, n_all: *mut u64, counts: *mut u64) -> c_int;
    pub fn psk_gather_sketches(comm: *mut PskComm, mine: *const *const PskSketch, n: u32, all: *mut *mut *mut PskSketch, counts: *mut u32) -> c_int;
    // learned-ANI regression: regression::get_model (lib.rs:614)
    pub fn psk_model_create(ctx: *mut PskCtx, nodes: *const PskTreeNode, n_nodes: u64, tree_first_node: *const u32,
                            n_trees: u32, bias: f32, shrinkage: f32, features: *const i32, n_features: u32,
                            out: *mut *mut PskModel) -> c_int;
    pub fn psk_model_load_json(ctx: *mut PskCtx, json: *const c_char, len: usize, out: *mut *mut PskModel) -> c_int;
    pub fn psk_model_load_file(ctx: *mut PskCtx, path: *const c_char, out: *mut *mut PskModel) -> c_int;
    pub fn psk_model_free(m: *mut PskModel);
    pub fn psk_model_info(m: *const PskModel, n_trees: *mut u32, n_nodes: *mut u64, n_features: *mut u32) -> c_int;
    pub fn psk_model_predict(m: *const PskModel, rows: *const f32, n_rows: u32, out: *mut f32) -> c_int;
    // database: markers.push + sketches.store (lib.rs:501-508)
    pub fn psk_ctx_small_query_stats(ctx: *mut PskCtx, taken: *mut u64, rerun: *mut u64, general: *mut u64) -> c_int;
    pub fn psk_pack2bit_host(src: *const u8, n: u64, dst: *mut u32, mode: c_int);
    pub fn psk_db_create(ctx: *mut PskCtx, p: *const PskParams, out: *mut *mut PskDb) -> c_int;
    pub fn psk_db_destroy(db: *mut PskDb);
    pub fn psk_db_add(db: *mut PskDb, name: *const c_char, s: *mut PskSketch) -> c_int;
    pub fn psk_db_add_batch(db: *mut PskDb, names: *const *const c_char, sketches: *const *mut PskSketch, n: u32) -> c_int;
    pub fn psk_db_size(db: *const PskDb) -> u32;
    pub fn psk_db_name(db: *const PskDb, index: u32) -> *const c_char;
    pub fn psk_db_sketch(db: *const PskDb, index: u32) -> *const PskSketch;
    // query: the allow_threads closure of lib.rs:569-659
    pub fn psk_query(db: *mut PskDb, q: *const PskSketch, o: *const PskQueryOpts,
                     hits: *mut *mut PskHit, n_hits: *mut u64) -> c_int;
    // the same from the caller's host bytes: the query is sketched (lib.rs:571, not stored) and queried in one call - a contig-sized
    // query runs as one launch sequence with one synchronisation
    pub fn psk_query_host(db: *mut PskDb, contigs: *const *const u8, lens: *const u64, n_contigs: u32, seed: c_int, o: *const PskQueryOpts,
                          hits: *mut *mut PskHit, n_hits: *mut u64) -> c_int;
    // many queries in one call (all-vs-all, bins of a metagenome): hits of query i are hits[offsets[i]..offsets[i+1]]
    pub fn psk_query_many(db: *mut PskDb, qs: *const *const PskSketch, n: u32, o: *const PskQueryOpts,
                          hits: *mut *mut PskHit, offsets: *mut u64) -> c_int;
    pub fn psk_query_many_min(db: *mut PskDb, qs: *const *const PskSketch, n: u32, o: *const PskQueryOpts,
                              hits: *mut *mut PskHitMin, offsets: *mut u64) -> c_int;
    // the two halves of `query`, for a disk-backed Database (markers resident, sketches loaded per query)
    pub fn psk_screen(db: *mut PskDb, q: *const PskSketch, screen_val: f64, rescue_small: c_int,
                      pass: *mut u8, shared: *mut u32) -> c_int;
    pub fn psk_chain(ctx: *mut PskCtx, refs: *const *const PskSketch, n: u32, q: *const PskSketch,
                     o: *const PskQueryOpts, out: *mut PskHit) -> c_int;
}

/// psk_status -> the PyErr pyskani already raises for that class of failure
pub fn check(status: c_int) -> pyo3::PyResult<()> {
    use pyo3::exceptions::*;
    if status == 0 { return Ok(()); }
    let msg = unsafe { std::ffi::CStr::from_ptr(psk_last_error()) }.to_string_lossy().into_owned();
    Err(match status {
        1 => PyValueError::new_err(msg),      // PSK_EINVAL   (k > 16 panics in skani today)
        2 => PyMemoryError::new_err(msg),     // PSK_ENOMEM
        5 => PyKeyError::new_err(msg),        // PSK_EKEY     (lib.rs:97,110)
        6 => PyOverflowError::new_err(msg),   // PSK_ELIMIT
        _ => PyRuntimeError::new_err(msg),    // PSK_EHIP, PSK_ENOMODEL, PSK_ERCCL
    })
}
