//! GPU execution engine for Ballista executors: implements `ExecutionEngine` / `QueryStageExecutor`
//! (ballista/executor/src/execution_engine.rs:34-60) on top of libgpuq.so's C ABI (include/gpuq.h).
//!
//! * `ffi`        -- the `extern "C"` declarations (bindgen-free, one to one with include/gpuq.h)
//! * `plan_walk`  -- DataFusion `ExecutionPlan` / `PhysicalExpr` tree -> the JSON mirror of `PhysicalPlanNode`
//!                   that `gpuq_plan_create` takes (downcast list of task_group.rs:143-167 and utils.rs:274-311)
//! * `engine`     -- `GpuExecutionEngine`, `GpuQueryStageExec`
//!
//! Status: source only; this image has no cargo.  See ../INTEGRATION.md.
pub mod engine;
pub mod ffi;
pub mod plan_walk;

pub use engine::{quiesce, GpuExecutionEngine, GpuQueryStageExec};
