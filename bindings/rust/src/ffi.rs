//! `extern "C"` surface of include/lpipm.h (only what the shim calls).
//! The crate denies `unsafe_code` (.cargo/config.toml:6); this module is the one scoped exception.
#![allow(unsafe_code, non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct lpipm_ctx {
    _private: [u8; 0],
}

/// include/lpipm.h `lpipm_opts` == InteriorPointBuilder fields (interior_point/mod.rs:41-48)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct lpipm_opts {
    pub tol: f64,
    pub alpha0: f64,
    pub max_iter: u64,
    pub ip: i32,
    pub solver_type: i32,
    pub disp: i32,
}

/// include/lpipm.h `lpipm_iter_row` (indicators.rs:8-23 + alpha)
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct lpipm_iter_row {
    pub alpha: f64,
    pub rho_p: f64,
    pub rho_d: f64,
    pub rho_a: f64,
    pub rho_g: f64,
    pub rho_mu: f64,
    pub obj: f64,
}

pub const LPIPM_OK: c_int = 0;
pub const LPIPM_UNCONSTRAINED: c_int = 1;
pub const LPIPM_NUMERICAL_PROBLEM: c_int = 2;
pub const LPIPM_INVALID_PARAMETER: c_int = 3;
pub const LPIPM_INCOMPATIBLE_DIMENSIONS: c_int = 4;
pub const LPIPM_INFEASIBLE: c_int = 5;
pub const LPIPM_UNBOUNDED: c_int = 6;
pub const LPIPM_ITERATION_LIMIT: c_int = 7;

extern "C" {
    pub fn lpipm_create(device: c_int, out: *mut *mut lpipm_ctx) -> c_int;
    pub fn lpipm_destroy(ctx: *mut lpipm_ctx);
    pub fn lpipm_upload(
        ctx: *mut lpipm_ctx, m: u64, n: u64, a: *const f64, lda: u64, b: *const f64, c: *const f64, c0: f64,
    ) -> c_int;
    pub fn lpipm_upload_slack(
        ctx: *mut lpipm_ctx, m: u64, n: u64, a: *const f64, lda: u64, b: *const f64, c: *const f64, c0: f64,
        n_slack: u64,
    ) -> c_int;
    pub fn lpipm_upload_ub_eq(
        ctx: *mut lpipm_ctx, n: u64, m_ub: u64, a_ub: *const f64, lda_ub: u64, b_ub: *const f64, m_eq: u64,
        a_eq: *const f64, lda_eq: u64, b_eq: *const f64, c: *const f64, c0: f64,
    ) -> c_int;
    pub fn lpipm_solve(
        ctx: *mut lpipm_ctx, opts: *const lpipm_opts, x_slack_out: *mut f64, fun_out: *mut f64,
        iterations_out: *mut u64, log: *mut lpipm_iter_row,
    ) -> c_int;
    /// InteriorPoint<f32> (src/float.rs:42-43): the same algorithm with every operation in f32; upload + solve in one call
    pub fn lpipm_solve_f32(
        ctx: *mut lpipm_ctx, m: u64, n: u64, a: *const f32, lda: u64, b: *const f32, c: *const f32, c0: f32,
        opts: *const lpipm_opts, x_slack_out: *mut f32, fun_out: *mut f32, iterations_out: *mut u64, log: *mut c_void,
    ) -> c_int;
    pub fn lpipm_strerror(status: c_int) -> *const c_char;
    pub fn lpipm_last_error_detail() -> *const c_char;

    // A batch of independent LPs on one device (BASELINE config 4): members of equal shape advance as
    // lockstep batches, one kernel launch covering all of them.
    pub fn lpipm_solve_batch(
        ctx: *mut lpipm_ctx, count: u64, m: *const u64, n: *const u64, a: *const *const f64,
        b: *const *const f64, c: *const *const f64, c0: *const f64, opts: *const lpipm_opts,
        x_slack_out: *const *mut f64, fun_out: *mut f64, iterations_out: *mut u64, status_out: *mut i32,
    ) -> c_int;
    pub fn lpipm_upload_lockstep(
        ctx: *mut lpipm_ctx, count: u64, m: u64, n: u64, a: *const *const f64, b: *const *const f64,
        c: *const *const f64, c0: *const f64,
    ) -> c_int;
    pub fn lpipm_solve_lockstep(
        ctx: *mut lpipm_ctx, opts: *const lpipm_opts, x_slack_out: *const *mut f64, fun_out: *mut f64,
        iterations_out: *mut u64, status_out: *mut i32,
    ) -> c_int;

    // One LP split by columns over ranks (BASELINE config 5): the caller supplies the all-reduce
    // (e.g. ncclAllReduce on `stream`); op 0 = sum, 1 = min.
    pub fn lpipm_set_collective(
        ctx: *mut lpipm_ctx, rank: c_int, world: c_int,
        f: Option<unsafe extern "C" fn(user: *mut c_void, dev_ptr: *mut c_void, count: u64, op: c_int, stream: *mut c_void) -> c_int>,
        user: *mut c_void,
    ) -> c_int;
    pub fn lpipm_upload_nsplit(
        ctx: *mut lpipm_ctx, m: u64, n_total: u64, n_local: u64, a_local: *const f64, lda: u64,
        b: *const f64, c_local: *const f64, c0: f64,
    ) -> c_int;
}
