//! FFI declarations + a safe wrapper for `libvrod_hip.so` (C ABI: `include/vrod.h`).
//!
//! This is the binding a vRod maintainer would add so that `SearchSimilarCommand::execute`
//! (reference `src/command/types.rs:121-132`, an empty stub) and `BulkInsertCommand::execute`
//! (`types.rs:69-80`) can call the MI355X scan.  Source only: there is no Rust toolchain in the
//! image this repository is built in, so this crate has never been compiled here.
#![allow(non_camel_case_types)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct vrod_index {
    _private: [u8; 0],
}

pub const VROD_DTYPE_F32: c_int = 0;
pub const VROD_DTYPE_BF16: c_int = 1;
pub const VROD_METRIC_COSINE: c_int = 0;
pub const VROD_METRIC_L2: c_int = 1;
pub const VROD_ID_NONE: u64 = u64::MAX;
pub const VROD_MAX_K: u32 = 3584;

#[repr(C)]
#[derive(Debug, Default, Clone, Copy)]
pub struct vrod_search_stats {
    pub path: u32,
    pub nq: u32,
    pub k: u32,
    pub kprime: u32,
    pub scan_launches: u32,
    pub fallback_queries: u32,
    pub scan_ms: f32,
    pub total_ms: f32,
    pub scan_bytes: f64,
    pub scan_flops: f64,
    pub max_fast_err: f32,
    pub eps_bound: f32,
    /// 1: the batched fast pass of an F32 handle ran on its bf16 [hi | lo] planes
    pub split_pass: u32,
    pub band_queries: u32,
    pub sample_ms: f32,
    pub exchange: u32,
    pub overlap_ms: f32,
}

extern "C" {
    pub fn vrod_index_create(out: *mut *mut vrod_index, dim: u32, dtype: c_int, metric: c_int,
                             device_ids: *const c_int, n_devices: c_int) -> c_int;
    pub fn vrod_index_destroy(idx: *mut vrod_index) -> c_int;
    pub fn vrod_index_reserve(idx: *mut vrod_index, n_rows: u64) -> c_int;
    pub fn vrod_index_add(idx: *mut vrod_index, rows: *const f32, n: u64) -> c_int;
    pub fn vrod_index_add_synthetic(idx: *mut vrod_index, seed: u64, first_row: u64, n: u64) -> c_int;
    pub fn vrod_index_count(idx: *const vrod_index, out_count: *mut u64) -> c_int;
    pub fn vrod_index_set_id_offset(idx: *mut vrod_index, offset: u64) -> c_int;
    pub fn vrod_index_get_rows(idx: *mut vrod_index, first: u64, n: u64, out_rows: *mut f32) -> c_int;
    pub fn vrod_search(idx: *mut vrod_index, queries: *const f32, nq: u32, k: u32,
                       out_ids: *mut u64, out_scores: *mut f32) -> c_int;
    pub fn vrod_search_device(idx: *mut vrod_index, d_queries: *const f32, nq: u32, k: u32,
                              d_out_ids: *mut u64, d_out_scores: *mut f32, stream: *mut c_void) -> c_int;
    pub fn vrod_search_synthetic_device(idx: *mut vrod_index, seed: u64, first_row: u64, nq: u32, k: u32,
                                        d_out_ids: *mut u64, d_out_scores: *mut f32, stream: *mut c_void) -> c_int;
    pub fn vrod_search_begin_device(idx: *mut vrod_index, d_queries: *const f32, nq: u32, k: u32,
                                    d_out_ids: *mut u64, d_out_scores: *mut f32, stream: *mut c_void) -> c_int;
    pub fn vrod_search_begin_synthetic_device(idx: *mut vrod_index, seed: u64, first_row: u64, nq: u32, k: u32,
                                              d_out_ids: *mut u64, d_out_scores: *mut f32,
                                              stream: *mut c_void) -> c_int;
    pub fn vrod_search_end(idx: *mut vrod_index) -> c_int;
    pub fn vrod_search_pending(idx: *const vrod_index, out_pending: *mut u32) -> c_int;
    pub fn vrod_merge_topk_device(device: c_int, metric: c_int, d_ids: *const u64, d_scores: *const f32,
                                  n_lists: u32, nq: u32, k: u32, d_out_ids: *mut u64,
                                  d_out_scores: *mut f32, stream: *mut c_void) -> c_int;
    pub fn vrod_merge_topk_packed_device(device: c_int, metric: c_int, d_packed: *const c_void, n_lists: u32,
                                         nq: u32, k: u32, d_out_ids: *mut u64, d_out_scores: *mut f32,
                                         stream: *mut c_void) -> c_int;
    pub fn vrod_index_set_path(idx: *mut vrod_index, path: c_int) -> c_int;
    pub fn vrod_index_set_profiling(idx: *mut vrod_index, on: c_int) -> c_int;
    pub fn vrod_index_last_stats(idx: *const vrod_index, out: *mut vrod_search_stats) -> c_int;
    pub fn vrod_index_shard_stats(idx: *const vrod_index, shard: u32, out_device: *mut c_int,
                                  out: *mut vrod_search_stats) -> c_int;
    pub fn vrod_last_error() -> *const c_char;
    pub fn vrod_version() -> *const c_char;
    pub fn vrod_synth_rows_device(device: c_int, seed: u64, first_row: u64, n: u64, dim: u32,
                                  d_out: *mut f32, stream: *mut c_void) -> c_int;
}

/// Joins the reference's `thiserror` enums (`src/main.rs:36-40`, `src/command/builder.rs:10-15`).
#[derive(Debug, thiserror::Error)]
pub enum ScanError {
    #[error("vrod_hip status {0}: {1}")]
    Device(i32, String),
    #[error("vector has {got} values, collection dimension is {want}")]
    Dim { got: usize, want: usize },
}

fn check(rc: c_int) -> Result<(), ScanError> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(vrod_last_error()) }.to_string_lossy().into_owned();
    Err(ScanError::Device(rc, msg))
}

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Metric { Cosine, L2 }
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Dtype { F32, Bf16 }

/// One collection's vectors in HBM.  `!Send + !Sync`, like the reference's `Rc<RefCell<Database>>`
/// (`src/command/types.rs:10`): calls on one handle are serialised by construction.
pub struct Collection {
    idx: *mut vrod_index,
    dim: usize,
}

impl Collection {
    pub fn new(dim: usize, dtype: Dtype, metric: Metric) -> Result<Self, ScanError> {
        let mut idx = std::ptr::null_mut();
        let dt = if dtype == Dtype::Bf16 { VROD_DTYPE_BF16 } else { VROD_DTYPE_F32 };
        let me = if metric == Metric::L2 { VROD_METRIC_L2 } else { VROD_METRIC_COSINE };
        check(unsafe { vrod_index_create(&mut idx, dim as u32, dt, me, std::ptr::null(), 0) })?;
        Ok(Self { idx, dim })
    }

    /// `embeddings` is the reference's own type (`src/utils/embeddings.rs:29`).
    pub fn add(&mut self, embeddings: &[Vec<f32>]) -> Result<(), ScanError> {
        let mut flat = Vec::with_capacity(embeddings.len() * self.dim);
        for e in embeddings {
            if e.len() != self.dim {
                return Err(ScanError::Dim { got: e.len(), want: self.dim });
            }
            flat.extend_from_slice(e);
        }
        check(unsafe { vrod_index_add(self.idx, flat.as_ptr(), embeddings.len() as u64) })
    }

    pub fn len(&self) -> u64 {
        let mut n = 0u64;
        unsafe { vrod_index_count(self.idx, &mut n) };
        n
    }

    /// Best-first `(ids, scores)`, `queries.len() * k` each; slots past `len()` are `(u64::MAX, NaN)`.
    pub fn search(&self, queries: &[Vec<f32>], k: usize) -> Result<(Vec<u64>, Vec<f32>), ScanError> {
        for q in queries {
            if q.len() != self.dim {
                return Err(ScanError::Dim { got: q.len(), want: self.dim });
            }
        }
        let flat: Vec<f32> = queries.iter().flatten().copied().collect();
        let mut ids = vec![0u64; queries.len() * k];
        let mut scores = vec![0f32; queries.len() * k];
        check(unsafe {
            vrod_search(self.idx, flat.as_ptr(), queries.len() as u32, k as u32, ids.as_mut_ptr(), scores.as_mut_ptr())
        })?;
        Ok((ids, scores))
    }
}

impl Drop for Collection {
    fn drop(&mut self) {
        unsafe { vrod_index_destroy(self.idx) };
    }
}

/// `f,f,...,f;word` -- the line format `write_embeddings_to_file` produces (`src/utils/embeddings.rs:55-61`).
pub fn parse_embedding_line(line: &str) -> Option<(Vec<f32>, &str)> {
    let (nums, word) = match line.find(';') {
        Some(i) => (&line[..i], &line[i + 1..]),
        None => (line, ""),
    };
    let v: Result<Vec<f32>, _> = nums.split(',').map(|t| t.trim().parse::<f32>()).collect();
    v.ok().filter(|v| !v.is_empty()).map(|v| (v, word))
}
