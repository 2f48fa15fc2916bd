//! `cfg(feature = "hip")` arm of `InteriorPoint::solve` -- the third backend next to the crate's
//! `cfg(feature = "blas")` fork (src/solvers/interior_point/newton_equations.rs:2-13).
//!
//! Lives at src/solvers/interior_point/hip_solver.rs in the reference tree; `mod.rs` gains
//!     #[cfg(feature = "hip")] mod hip_solver;
//! and, under `feature = "hip"`, the generic `impl<F: Float> Solver<F> for InteriorPoint<F>` (mod.rs:161-169) is
//! REPLACED BY TWO CONCRETE IMPLS (a `cfg` cannot remove one instantiation of a generic impl, and a generic impl
//! beside a concrete one is E0119): `impl Solver<f64> for InteriorPoint<f64>` and `impl Solver<f32> for InteriorPoint<f32>`,
//! both below, both through the FFI (`lpipm_solve` / `lpipm_solve_f32`: the f32 instantiation runs on generic,
//! scalar-type-templated HIP kernels with every operation in f32).  The unchanged CPU body moves into the inherent method
//! `InteriorPoint::<F>::solve_cpu` (what the crate uses without the feature).  The exact edit: ../interior_point_mod.rs.patch.
#![allow(unsafe_code)]

use ndarray::Array1;

use crate::error::LinearProgramError;
use crate::ffi::*;
use crate::linear_program::Problem;
use crate::solvers::{OptimizeResult, Solver};

use super::{EquationSolverType, InteriorPoint};

struct Ctx(*mut lpipm_ctx);
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { lpipm_destroy(self.0) }
    }
}

fn to_error(status: i32, x: Option<Array1<f64>>) -> LinearProgramError<f64> {
    match status {
        LPIPM_UNCONSTRAINED => LinearProgramError::Unconstrained,
        LPIPM_INVALID_PARAMETER => LinearProgramError::InvalidParameter("rejected by the HIP backend"),
        LPIPM_INCOMPATIBLE_DIMENSIONS => LinearProgramError::IncompatibleInputDimensions,
        LPIPM_INFEASIBLE => LinearProgramError::Infeasible,
        LPIPM_UNBOUNDED => LinearProgramError::Unbounded,
        LPIPM_ITERATION_LIMIT => LinearProgramError::IterationLimitExceeded(x.unwrap_or_else(|| Array1::zeros(0))),
        // LPIPM_NUMERICAL_PROBLEM and every runtime failure (>= 100: no analogue in error.rs).
        // Documented divergence: a HIP/driver error surfaces as NumericalProblem, never as a wrong x.
        _ => LinearProgramError::NumericalProblem,
    }
}

impl Solver<f64> for InteriorPoint<f64> {
    /// Same contract as interior_point/mod.rs:161-168: stateless, re-entrant, never panics on a
    /// numerical failure.  One lpipm_ctx per call (stream + device buffers); A is uploaded once.
    fn solve(&self, problem: &Problem<f64>) -> Result<OptimizeResult<f64>, LinearProgramError<f64>> {
        let a = problem.A().as_standard_layout(); // row-major, as ProblemBuilder::build leaves it (linear_program.rs:145-156)
        let (m, n) = a.dim();
        let b = problem.b().as_standard_layout();
        let c = problem.c().as_standard_layout();
        let opts = lpipm_opts {
            tol: self.tol,
            alpha0: self.alpha0,
            max_iter: self.max_iter as u64,
            ip: self.ip as i32,
            solver_type: match self.solver_type {
                EquationSolverType::Cholesky => 0,
                EquationSolverType::Inverse => 1,
                EquationSolverType::LeastSquares => 2,
            },
            disp: self.disp as i32,
        };
        let mut raw: *mut lpipm_ctx = std::ptr::null_mut();
        let rc = unsafe { lpipm_create(0, &mut raw) };
        if rc != LPIPM_OK {
            return Err(to_error(rc, None));
        }
        let ctx = Ctx(raw);
        // n_slack: the last columns are the [I; 0] slack block (linear_program.rs:147-161); the backend
        // then neither copies nor multiplies them
        let rc = unsafe {
            lpipm_upload_slack(ctx.0, m as u64, n as u64, a.as_ptr(), n as u64, b.as_ptr(), c.as_ptr(),
                               problem.c0(), problem.n_slack() as u64)
        };
        if rc != LPIPM_OK {
            return Err(to_error(rc, None));
        }
        let mut x_slack = Array1::<f64>::zeros(n);
        let (mut fun, mut iteration) = (0.0f64, 0u64);
        let rc = unsafe {
            lpipm_solve(ctx.0, &opts, x_slack.as_mut_ptr(), &mut fun, &mut iteration, std::ptr::null_mut())
        };
        match rc {
            LPIPM_OK => {
                // mod.rs:165-167: fun = c.x + c0 (computed on device from the same x), drop the slack tail
                let x = problem.denormalize_x_into(x_slack);
                Ok(OptimizeResult::new(x, fun, iteration as usize))
            }
            LPIPM_ITERATION_LIMIT => Err(to_error(rc, Some(x_slack))), // mod.rs:237-239
            _ => Err(to_error(rc, None)),
        }
    }
}

fn to_error_f32(status: i32, x: Option<Array1<f32>>) -> LinearProgramError<f32> {
    match status {
        LPIPM_UNCONSTRAINED => LinearProgramError::Unconstrained,
        LPIPM_INVALID_PARAMETER => LinearProgramError::InvalidParameter("rejected by the HIP backend"),
        LPIPM_INCOMPATIBLE_DIMENSIONS => LinearProgramError::IncompatibleInputDimensions,
        LPIPM_INFEASIBLE => LinearProgramError::Infeasible,
        LPIPM_UNBOUNDED => LinearProgramError::Unbounded,
        LPIPM_ITERATION_LIMIT => LinearProgramError::IterationLimitExceeded(x.unwrap_or_else(|| Array1::zeros(0))),
        _ => LinearProgramError::NumericalProblem,
    }
}

impl Solver<f32> for InteriorPoint<f32> {
    /// interior_point/mod.rs:161-168 for F = f32: `lpipm_solve_f32` uploads, solves and releases in one call.  Only the
    /// Cholesky arm exists in f32 on the device; the other arms fall back to the crate's own CPU body.
    fn solve(&self, problem: &Problem<f32>) -> Result<OptimizeResult<f32>, LinearProgramError<f32>> {
        if !matches!(self.solver_type, EquationSolverType::Cholesky) {
            return self.solve_cpu(problem);
        }
        let a = problem.A().as_standard_layout();
        let (m, n) = a.dim();
        let b = problem.b().as_standard_layout();
        let c = problem.c().as_standard_layout();
        let opts = lpipm_opts {
            tol: self.tol as f64, alpha0: self.alpha0 as f64, max_iter: self.max_iter as u64, ip: self.ip as i32,
            solver_type: 0, disp: self.disp as i32,
        };
        let mut raw: *mut lpipm_ctx = std::ptr::null_mut();
        let rc = unsafe { lpipm_create(0, &mut raw) };
        if rc != LPIPM_OK {
            return Err(to_error_f32(rc, None));
        }
        let ctx = Ctx(raw);
        let mut x_slack = Array1::<f32>::zeros(n);
        let (mut fun, mut iteration) = (0.0f32, 0u64);
        let rc = unsafe {
            lpipm_solve_f32(ctx.0, m as u64, n as u64, a.as_ptr(), n as u64, b.as_ptr(), c.as_ptr(), problem.c0(), &opts,
                            x_slack.as_mut_ptr(), &mut fun, &mut iteration, std::ptr::null_mut())
        };
        match rc {
            LPIPM_OK => Ok(OptimizeResult::new(problem.denormalize_x_into(x_slack), fun, iteration as usize)),
            LPIPM_ITERATION_LIMIT => Err(to_error_f32(rc, Some(x_slack))),
            _ => Err(to_error_f32(rc, None)),
        }
    }
}
