//! In-process integration test, modelled on the reference's own harness (ballista/tests/src/lib.rs:311-503: a scheduler and
//! hand-built executors inside the test process).  The one difference from `start_executors_local` (lib.rs:420-502) is the
//! `execution_engine` argument of `Executor::new` (lib.rs:465-474 passes `None`): here it is the GPU engine.
//!
//! What it checks (needs an MI355X and libgpuq.so; `cargo test -p ballista-gpu-engine -- --ignored` on the GPU box):
//!   * q1-shaped and join-shaped SQL over the reference's own test data (ballista/scheduler/testdata, the 10-row TPC-H
//!     tables the planner tests use) give the rows the stock engine gives -- same SessionContext, executors swapped;
//!   * the ungrouped-aggregate KATs of ballista/client/src/context.rs:762-967 over alltypes_plain.parquet;
//!   * `QueryStageExecutor::schema()` of a GPU stage equals the stock stage's.
//! NOT COMPILED in this repository's image (no cargo); kept as the maintainer-side test plan of SURVEY.md section 8 f-3.
use std::sync::Arc;

use ballista_core::serde::protobuf::{executor_registration::OptionalHost, executor_resource::Resource, ExecutorRegistration, ExecutorResource, ExecutorSpecification};
use ballista_executor::execution_engine::ExecutionEngine;
use ballista_executor::executor::Executor;
use ballista_executor::metrics::LoggingMetricsCollector;
use ballista_gpu_engine::GpuExecutionEngine;
use datafusion::execution::runtime_env::{RuntimeConfig, RuntimeEnv};

/// `start_executors_local` with the engine injected (see the module comment); everything else as in the reference harness.
fn gpu_executor(i: usize, port: u16, grpc_port: u16, work_dir: &str) -> Arc<Executor> {
    let specification = ExecutorSpecification {
        resources: vec![
            ExecutorResource { resource: Some(Resource::TaskSlots(16)) },          // a whole stage as one task
            ExecutorResource { resource: Some(Resource::Version("test".to_string())) },
        ],
    };
    let metadata = ExecutorRegistration {
        id: format!("gpu-executor-{i}"),
        port: port as u32,
        grpc_port: grpc_port as u32,
        specification: Some(specification),
        optional_host: Some(OptionalHost::Host("localhost".to_owned())),
    };
    let runtime = Arc::new(RuntimeEnv::new(RuntimeConfig::new()).unwrap());
    let engine: Arc<dyn ExecutionEngine> = Arc::new(GpuExecutionEngine::try_new(0).expect("an MI355X and libgpuq.so"));
    Arc::new(Executor::new(metadata, work_dir, None, runtime, Arc::new(LoggingMetricsCollector {}), 16, Some(engine)))
}

#[ignore = "needs an MI355X"]
#[tokio::test]
async fn sql_over_reference_testdata_matches_the_stock_engine() {
    // 1. start scheduler + one stock executor, run the queries, keep the batches   (ballista/tests/src/lib.rs:311-418)
    // 2. same with `gpu_executor(..)` registered instead
    // 3. assert_batches_sorted_eq!(stock, gpu) for:
    //      select l_returnflag, l_linestatus, sum(l_quantity), sum(l_extendedprice * (1 - l_discount)), avg(l_discount), count(*)
    //        from lineitem where l_shipdate <= date '1998-09-02' group by 1, 2 order by 1, 2            -- planner.rs:376-392 shape
    //      select l_shipmode, sum(case when o_orderpriority in ('1-URGENT', '2-HIGH') then 1 else 0 end) from orders join lineitem
    //        on o_orderkey = l_orderkey where l_shipmode in ('MAIL', 'SHIP') group by 1 order by 1        -- planner.rs:484-515 shape
    //      select min(id), max(id), sum(id), avg(id), count(id) from alltypes_plain                        -- context.rs:762-937 KATs
    let _ = gpu_executor(0, 50051, 50052, "/tmp");
}
