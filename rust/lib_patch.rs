// rust/lib_patch.rs — the bodies that replace skani calls in src/pyskani/_skani/lib.rs when pyskani is bound to
// libpyskani_amd.so. Each item names the reference lines it replaces. NOT compiled in this
// repository (no Rust toolchain); it is source a maintainer drops into lib.rs next to `mod ffi; mod model;`.
use crate::ffi::{self, PskCtx, PskDb, PskHit, PskModel, PskParams, PskQueryOpts, PskSketch};
use pyo3::prelude::*;
use std::ffi::{CStr, CString};
use std::sync::Mutex;

/// replaces `pub struct Sketch(skani::types::Sketch)` (sketch.rs:4-8): the handle owns a device-resident sketch
pub struct GpuSketch { pub handle: *mut PskSketch, pub name: String }
unsafe impl Send for GpuSketch {}
unsafe impl Sync for GpuSketch {}
impl Drop for GpuSketch { fn drop(&mut self) { if !self.handle.is_null() { unsafe { ffi::psk_sketch_free(self.handle) } } } }

/// added to `struct Database` (lib.rs:132-137) beside `params`; `markers` and the Memory arm of `sketches` move into `gpu`
pub struct GpuState {
    pub ctx: *mut PskCtx,
    pub db: *mut PskDb,
    pub params: PskParams,
    /// one uploaded model per `learned` flag, built on first use (lib.rs:611-614 asks for it on every query)
    pub models: Mutex<[Option<usize>; 2]>,
}
unsafe impl Send for GpuState {}
unsafe impl Sync for GpuState {}

impl GpuState {
    /// in `Database::__init__` (lib.rs:413-417), after `SketchParams::new`
    pub fn new(c: usize, marker_c: usize, k: usize) -> PyResult<Self> {
        let params = PskParams { c: c as i32, marker_c: marker_c as i32, k: k as i32 };
        let (mut ctx, mut db) = (std::ptr::null_mut(), std::ptr::null_mut());
        unsafe {
            ffi::check(ffi::psk_ctx_create(0, &mut ctx))?;
            ffi::check(ffi::psk_db_create(ctx, &params, &mut db))?;
        }
        Ok(GpuState { ctx, db, params, models: Mutex::new([None, None]) })
    }

    /// replaces the body of `Database::_sketch` (lib.rs:140-185): `Sketch::new`, the `MIN_LENGTH_CONTIG` filter (:156), the
    /// `{name}_{i}` bookkeeping (:157-161) and the per-contig `fmh_seeds` loop (:165-171) all happen inside the library
    pub fn sketch<'c, C>(&self, name: String, contigs: C, seed: bool) -> PyResult<GpuSketch>
    where C: IntoIterator<Item = &'c [u8]> {
        let views: Vec<&[u8]> = contigs.into_iter().collect();
        let ptrs: Vec<*const u8> = views.iter().map(|v| v.as_ptr()).collect();
        let lens: Vec<u64> = views.iter().map(|v| v.len() as u64).collect();
        let mut handle = std::ptr::null_mut();
        unsafe {
            ffi::check(ffi::psk_sketch_host(self.ctx, &self.params, ptrs.as_ptr(), lens.as_ptr(), ptrs.len() as u32,
                                            seed as i32, &mut handle))?;
        }
        Ok(GpuSketch { handle, name })
    }

    /// replaces `get_markers_only` + `markers.push(marker)` + `sketches.store(sketch)` (lib.rs:495-508); ownership of the
    /// sketch moves into the database exactly as `store` moves it today
    pub fn add(&self, mut sketch: GpuSketch) -> PyResult<()> {
        let cname = CString::new(sketch.name.clone()).map_err(|e| pyo3::exceptions::PyValueError::new_err(e.to_string()))?;
        let h = std::mem::replace(&mut sketch.handle, std::ptr::null_mut());
        unsafe { ffi::check(ffi::psk_db_add(self.db, cname.as_ptr(), h)) }
    }

    fn model_for(&self, learned: bool) -> PyResult<*const PskModel> {
        let mut slot = self.models.lock().map_err(|_| pyo3::exceptions::PyRuntimeError::new_err("Poisoned lock"))?;
        let i = learned as usize;
        if slot[i].is_none() {
            let m = unsafe { crate::model::upload(self.ctx, self.params.c as usize, learned)? };
            slot[i] = Some(m as usize);
        }
        Ok(slot[i].unwrap() as *const PskModel)
    }

    /// replaces the whole `py.allow_threads(move || { .. })` closure of `Database::query` (lib.rs:569-659): the CommandParams
    /// literal (:573-601), screen_val (:603-609), use_learned_ani / get_model (:611-614), the check_markers_quickly loop
    /// (:617-637), map_params_from_sketch + chain_seeds (:646-653) and the `ani > 0.1` filter (:654). Call it inside
    /// `allow_threads`, as today.
    #[allow(clippy::too_many_arguments)]
    pub fn query<'c, C>(&self, name: String, contigs: C, seed: bool, learned_ani: Option<bool>, median: bool, robust: bool,
                        cutoff: Option<f64>, faster_small: bool) -> PyResult<Vec<skani::types::AniEstResult>>
    where C: IntoIterator<Item = &'c [u8]> {
        // lib.rs:571: the query sketch is not stored - psk_query_host sketches and queries in one call
        let views: Vec<&[u8]> = contigs.into_iter().collect();
        let ptrs: Vec<*const u8> = views.iter().map(|v| v.as_ptr()).collect();
        let lens: Vec<u64> = views.iter().map(|v| v.len() as u64).collect();
        let learned = learned_ani.unwrap_or_else(|| skani::regression::use_learned_ani(self.params.c as usize, false, false, median));
        let model = self.model_for(learned)?;                         // null when get_model returns None
        let opts = PskQueryOpts {
            learned_ani: if model.is_null() { 0 } else { learned as i32 },
            median: median as i32, robust: robust as i32, faster_small: faster_small as i32,
            cutoff: cutoff.unwrap_or(0.0),                            // 0 = SEARCH_ANI_CUTOFF_DEFAULT inside the library
            min_aligned_frac: 0.0,                                    // 0 = D_FRAC_COVER_CUTOFF / 100
            model,
        };
        let (mut hits, mut n): (*mut PskHit, u64) = (std::ptr::null_mut(), 0);
        unsafe { ffi::check(ffi::psk_query_host(self.db, ptrs.as_ptr(), lens.as_ptr(), ptrs.len() as u32, seed as i32, &opts, &mut hits, &mut n))?; }
        let mut out = Vec::with_capacity(n as usize);
        for i in 0..n as usize {
            let h = unsafe { *hits.add(i) };
            let ref_name = unsafe { CStr::from_ptr(ffi::psk_db_name(self.db, h.ref_index)) }.to_string_lossy().into_owned();
            out.push(skani::types::AniEstResult {                     // the fields Hit::from reads, hit.rs:119-123
                ani: h.ani, align_fraction_query: h.af_query, align_fraction_ref: h.af_ref,
                ref_file: ref_name, query_file: name.clone(), ..Default::default()
            });
        }
        unsafe { ffi::psk_free(hits as *mut _) };
        Ok(out)
    }
}

impl Drop for GpuState {
    fn drop(&mut self) {
        unsafe {
            for m in self.models.lock().unwrap().iter().flatten() { ffi::psk_model_free(*m as *mut PskModel) }
            ffi::psk_db_destroy(self.db);
            ffi::psk_ctx_destroy(self.ctx);
        }
    }
}

// In lib.rs itself the edits are then three lines:
//   sketch():  let sketch = py.allow_threads(|| self.gpu._sketch(name, views, seed))?;  self.gpu.add(sketch)?;      (:493-508)
//   query():   py.allow_threads(move || Ok(self.gpu.query(name, views, seed, learned_ani, median, robust, cutoff,
//              faster_small)?.into_iter().map(Hit::from).collect()))                                                  (:569-659)
//   __init__:  gpu: GpuState::new(compression, marker_compression, k)?                                               (:413-417)
