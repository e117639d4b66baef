//! `ballista-gpu-executor`: the stock executor binary (ballista/executor/src/bin/main.rs) with ONE difference -- the
//! `execution_engine` field of `ExecutorProcessConfig` (executor_process.rs:92-94), `None` in the stock binary (bin/main.rs:81),
//! carries the GPU engine.  Run one process per GPU (GPUQ_DEVICE = 0..7) with `--concurrent-tasks` >= `ballista.shuffle.partitions`
//! so that a whole stage arrives as one task.
use std::sync::Arc;

use ballista_executor::executor_process::{start_executor_process, ExecutorProcessConfig};
use ballista_gpu_engine::GpuExecutionEngine;

// The option struct and its parsing are the stock binary's: this crate's build.rs runs configure_me_codegen over the SAME
// executor_config_spec.toml ([package.metadata.configure_me.bin] executor = ... in Cargo.toml), which writes
// OUT_DIR/executor_configure_me_config.rs for THIS crate (ballista/executor/src/bin/main.rs:30-35 includes its own copy the same way).
#[macro_use]
extern crate configure_me;

#[allow(clippy::all, warnings)]
mod config {
    include!(concat!(env!("OUT_DIR"), "/executor_configure_me_config.rs"));
}
use config::prelude::*;

#[tokio::main]
async fn main() -> anyhow::Result<()> {
    let (opt, _remaining_args) = Config::including_optional_config_files(&["/etc/ballista/executor.toml"]).unwrap_or_exit();
    let device: i32 = std::env::var("GPUQ_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
    let engine = GpuExecutionEngine::try_new(device).map_err(|e| anyhow::anyhow!("{e}"))?;
    let log_file_name_prefix = format!("executor_{}_{}", opt.external_host.clone().unwrap_or_else(|| "localhost".to_string()), opt.bind_port);
    let config = ExecutorProcessConfig {
        special_mod_log_level: opt.log_level_setting,
        external_host: opt.external_host,
        bind_host: opt.bind_host,
        port: opt.bind_port,
        grpc_port: opt.bind_grpc_port,
        version: opt.executor_version,
        scheduler_host: opt.scheduler_host,
        scheduler_port: opt.scheduler_port,
        scheduler_connect_timeout_seconds: opt.scheduler_connect_timeout_seconds,
        concurrent_tasks: opt.concurrent_tasks,
        task_scheduling_policy: opt.task_scheduling_policy,
        work_dir: opt.work_dir,
        log_dir: opt.log_dir,
        log_file_name_prefix,
        log_rotation_policy: opt.log_rotation_policy,
        print_thread_info: opt.print_thread_info,
        job_data_ttl_seconds: opt.job_data_ttl_seconds,
        job_data_clean_up_interval_seconds: opt.job_data_clean_up_interval_seconds,
        execution_engine: Some(Arc::new(engine)),          // <- the one change
        replication_url: opt.replication_url,
    };
    let r = start_executor_process(config).await;
    // background specialisations may still be inside hiprtc: let them finish before the process' static destructors run
    // (include/gpuq.h gpuq_jit_quiesce; DESIGN.md section 7 "Exit order")
    ballista_gpu_engine::quiesce();
    r
}
