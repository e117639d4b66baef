//! `ExecutionEngine` / `QueryStageExecutor` over libgpuq (ballista/executor/src/execution_engine.rs:34-60).
use std::ffi::CString;
use std::fmt::{Debug, Display, Formatter};
use std::os::raw::c_void;
use std::sync::{Arc, Mutex};

use arrow::array::{Array, StringArray, StructArray, UInt32Array, UInt64Array};
use arrow::datatypes::{DataType, Field, Schema, SchemaRef};
use arrow::ffi::{from_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use arrow::record_batch::RecordBatch;
use async_trait::async_trait;
use ballista_core::execution_plans::ShuffleWriterExec;
use ballista_core::replicator;
use ballista_core::serde::protobuf::ShuffleWritePartition;
use ballista_executor::execution_engine::{DefaultExecutionEngine, ExecutionEngine, QueryStageExecutor};
use datafusion::common::{DataFusionError, Result};
use datafusion::execution::context::TaskContext;
use datafusion::physical_plan::metrics::{Count, MetricBuilder, MetricsSet, ExecutionPlanMetricsSet, Time};
use datafusion::physical_plan::ExecutionPlan;
use futures::StreamExt;
use tokio::sync::mpsc;

use crate::ffi::*;
use crate::plan_walk::{walk_stage, WalkedPlan};

/// One device context per executor process (one executor per GPU: INTEGRATION.md section 5).
struct Ctx(*mut gpuq_ctx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}
impl Drop for Ctx {
    // background specialisations may still be inside hiprtc: they finish before the context (and, at process exit, the library's
    // statics) go away (include/gpuq.h gpuq_jit_quiesce)
    fn drop(&mut self) { unsafe { gpuq_jit_quiesce(); gpuq_ctx_free(self.0) } }
}

/// Waits for the library's background compiles; a custom `main` calls it before it returns (bin/gpu_executor.rs).
pub fn quiesce() { unsafe { gpuq_jit_quiesce() } }

pub struct GpuExecutionEngine {
    ctx: Arc<Ctx>,
    /// stages the device path refuses as a whole (no supported node above the leaves) go to the stock engine
    fallback: DefaultExecutionEngine,
}

impl GpuExecutionEngine {
    pub fn try_new(device_ordinal: i32) -> Result<Self> {
        let ctx = unsafe { gpuq_ctx_create(device_ordinal, std::ptr::null()) };
        if ctx.is_null() {
            let why = unsafe { std::ffi::CStr::from_ptr(gpuq_last_error(std::ptr::null_mut())).to_string_lossy().into_owned() };
            return Err(DataFusionError::Execution(format!("gpuq: {why}")));
        }
        Ok(Self { ctx: Arc::new(Ctx(ctx)), fallback: DefaultExecutionEngine {} })
    }

    /// The device-memory budget of this executor's tasks: what `RuntimeConfig::with_memory_limit` is to the stock engine.  A task that
    /// would cross it fails with ResourcesExhausted (GPUQ_ERR_CAPACITY); the executor goes on.  0 = no limit.
    pub fn with_memory_limit(self, bytes: usize) -> Self {
        unsafe { gpuq_memory_limit(bytes as i64) };
        self
    }
}

impl ExecutionEngine for GpuExecutionEngine {
    fn create_query_stage_exec(
        &self,
        job_id: String,
        stage_id: usize,
        plan: Arc<dyn ExecutionPlan>,
        work_dir: &str,
        sender: Option<mpsc::Sender<replicator::Command>>,
    ) -> Result<Arc<dyn QueryStageExecutor>> {
        // the plan the scheduler sends always starts with a ShuffleWriterExec whose work_dir is "" (serde/mod.rs:191)
        let writer = plan.as_any().downcast_ref::<ShuffleWriterExec>().ok_or_else(|| {
            DataFusionError::Internal("Plan passed to create_query_stage_exec is not a ShuffleWriterExec".to_string())
        })?;
        let walked = walk_stage(writer, &job_id, stage_id, work_dir)?;
        // nothing but host leaves under the writer: the device would only re-encode batches -- leave the stage to DataFusion
        if walked.json["ShuffleWriterExec"]["input"].get("MemoryExec").is_some() {
            return self.fallback.create_query_stage_exec(job_id, stage_id, plan, work_dir, sender);
        }
        let text = CString::new(walked.json.to_string()).map_err(|e| DataFusionError::Internal(e.to_string()))?;
        let mut handle: *mut gpuq_plan = std::ptr::null_mut();
        let rc = unsafe { gpuq_plan_create(self.ctx.0, text.as_ptr(), &mut handle) };
        if rc != GPUQ_OK {
            return Err(DataFusionError::Execution(format!("gpuq_plan_create: {}", plan_error())));
        }
        Ok(Arc::new(GpuQueryStageExec {
            ctx: self.ctx.clone(),
            plan: Mutex::new(PlanHandle(handle)),
            walked,
            stage_schema: plan.children()[0].schema(),
            partitions: writer.partitions().to_vec(),
            sender,
            metrics: ExecutionPlanMetricsSet::new(),
            display: format!("GpuQueryStageExec: job={job_id} stage={stage_id}"),
            job_id,
        }))
    }
}

struct PlanHandle(*mut gpuq_plan);
unsafe impl Send for PlanHandle {}
impl Drop for PlanHandle {
    fn drop(&mut self) { unsafe { gpuq_plan_free(self.0) } }
}

pub struct GpuQueryStageExec {
    ctx: Arc<Ctx>,
    plan: Mutex<PlanHandle>,          // one task per plan at a time (include/gpuq.h)
    walked: WalkedPlan,
    stage_schema: SchemaRef,
    partitions: Vec<usize>,
    sender: Option<mpsc::Sender<replicator::Command>>,
    metrics: ExecutionPlanMetricsSet,
    display: String,
    job_id: String,                   // replicator::Command::Replicate carries it (shuffle_writer.rs:429-447)
}

impl Debug for GpuQueryStageExec {
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result { write!(f, "{}", self.display) }
}
impl Display for GpuQueryStageExec {
    // the metrics log prints this (metrics/mod.rs:46-57)
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        let mut buf = vec![0u8; 1 << 16];
        let plan = self.plan.lock().unwrap();
        let rc = unsafe { gpuq_plan_metrics(plan.0, buf.as_mut_ptr() as *mut _, buf.len()) };
        let m = if rc == GPUQ_OK { String::from_utf8_lossy(&buf[..buf.iter().position(|b| *b == 0).unwrap_or(0)]).into_owned() } else { String::new() };
        write!(f, "{} metrics={}", self.display, m)
    }
}

/// A running device task; dropping it (the executor drops the future to cancel, executor.rs:201-240) cancels and joins.
struct TaskGuard(*mut gpuq_task);
unsafe impl Send for TaskGuard {}
impl Drop for TaskGuard {
    fn drop(&mut self) { unsafe { gpuq_task_free(self.0) } }
}

/// Device tables imported for the host leaves of one task; freed after the task.
struct Imported(Vec<*mut gpuq_table>);
unsafe impl Send for Imported {}
impl Drop for Imported {
    fn drop(&mut self) { for t in &self.0 { unsafe { gpuq_table_free(*t) } } }
}

impl GpuQueryStageExec {
    /// Run one host leaf partition with DataFusion and import its batches (concatenated) as one device table.
    async fn import_leaf(&self, leaf: &Arc<dyn ExecutionPlan>, partition: usize, context: Arc<TaskContext>) -> Result<*mut gpuq_table> {
        let mut stream = leaf.execute(partition, context)?;
        let mut batches = vec![];
        while let Some(b) = stream.next().await { batches.push(b?); }
        let batch = arrow::compute::concat_batches(&leaf.schema(), &batches)?;
        let sa: StructArray = batch.into();
        let (ffi_array, ffi_schema) = arrow::ffi::to_ffi(&sa.to_data())?;
        let mut out: *mut gpuq_table = std::ptr::null_mut();
        let rc = unsafe { gpuq_table_import_arrow(self.ctx.0, std::ptr::null_mut(), &ffi_array, &ffi_schema, &mut out) };
        if rc != GPUQ_OK {
            let why = unsafe { std::ffi::CStr::from_ptr(gpuq_last_error(self.ctx.0)).to_string_lossy().into_owned() };
            return Err(DataFusionError::Execution(format!("gpuq_table_import_arrow: {why}")));
        }
        Ok(out)
    }
}

#[async_trait]
impl QueryStageExecutor for GpuQueryStageExec {
    async fn execute_query_stage(&self, _input_partitions: Vec<usize>, context: Arc<TaskContext>) -> Result<Vec<ShuffleWritePartition>> {
        // as DefaultQueryStageExec: the partition list lives in the plan; the writer runs partition 0
        // 1. host leaves -> device tables
        let mut imported = Imported(vec![]);
        for (leaf, part) in &self.walked.host_leaves {
            imported.0.push(self.import_leaf(leaf, *part, context.clone()).await?);
        }
        // 2. start the plan on the library's worker thread; await it without blocking the task-runner's worker
        //    (cpu_bound_executor.rs:94-131: blocking in poll stalls a worker).  The column / input arrays hold raw pointers (!Send):
        //    they live in this block only -- gpuq_plan_execute_async copies them -- so that nothing !Send is held across an await
        //    (QueryStageExecutor is #[async_trait]: its future must be Send).
        let task = {
            let mut cols: Vec<Vec<gpuq_column>> = vec![];
            let mut inputs: Vec<gpuq_input> = vec![];
            for t in &imported.0 {
                let nc = unsafe { gpuq_table_num_columns(*t) };
                let mut v = vec![unsafe { std::mem::zeroed::<gpuq_column>() }; nc as usize];
                for i in 0..nc { unsafe { gpuq_table_column(*t, i, &mut v[i as usize], std::ptr::null_mut()) }; }
                cols.push(v);
            }
            for (t, v) in imported.0.iter().zip(cols.iter()) {
                inputs.push(gpuq_input { cols: v.as_ptr(), n_cols: v.len() as i32, n_via: 0, n_rows: unsafe { gpuq_table_num_rows(*t) }, via: [std::ptr::null(); 3], n_rows_dev: std::ptr::null() });
            }
            let plan = self.plan.lock().unwrap();
            let mut t: *mut gpuq_task = std::ptr::null_mut();
            let rc = unsafe { gpuq_plan_execute_async(plan.0, std::ptr::null_mut::<c_void>(), 0, inputs.as_ptr(), inputs.len() as i32, &mut t) };
            if rc != GPUQ_OK { return Err(DataFusionError::Execution(format!("gpuq_plan_execute_async: {}", plan_error()))); }
            TaskGuard(t)
        };
        loop {
            let mut done = 0;
            unsafe { gpuq_task_poll(task.0, &mut done) };
            if done != 0 { break; }
            tokio::time::sleep(std::time::Duration::from_micros(200)).await;      // an await point: dropping the future here cancels (TaskGuard::drop)
        }
        // 3. the result batch (partition, path, num_rows, num_batches, num_bytes) -> Vec<ShuffleWritePartition>; the raw result
        //    pointer is converted and freed here, before the next await (sender.send below)
        let batch = {
            let mut res: *mut gpuq_result = std::ptr::null_mut();
            let rc = unsafe { gpuq_task_wait(task.0, &mut res) };
            if rc != GPUQ_OK {
                let why = plan_error();
                // "FetchFailed: ..." must reach the scheduler as such: it re-runs the map stage (shuffle_reader.rs:654)
                return Err(if why.starts_with("FetchFailed") { DataFusionError::Execution(why) } else { DataFusionError::Execution(format!("gpuq: {why}")) });
            }
            let b = export_result(self.ctx.0, res);
            unsafe { gpuq_result_free(res) };
            b?
        };
        let part = batch.column(0).as_any().downcast_ref::<UInt32Array>().unwrap();
        let path = batch.column(1).as_any().downcast_ref::<StringArray>().unwrap();
        let rows = batch.column(2).as_any().downcast_ref::<UInt64Array>().unwrap();
        let nb = batch.column(3).as_any().downcast_ref::<UInt64Array>().unwrap();
        let bytes = batch.column(4).as_any().downcast_ref::<UInt64Array>().unwrap();
        let mut out = vec![];
        for i in 0..batch.num_rows() {
            out.push(ShuffleWritePartition {
                partitions: self.partitions.iter().map(|p| *p as u32).collect(),      // shuffle_writer.rs:411-420
                output_partition: part.value(i),
                path: path.value(i).to_string(),
                num_batches: nb.value(i),
                num_rows: rows.value(i),
                num_bytes: bytes.value(i),
            });
            if rows.value(i) > 0 {
                if let Some(sender) = self.sender.as_ref() {      // replication hand-off, shuffle_writer.rs:429-447
                    let cmd = replicator::Command::Replicate { job_id: self.job_id.clone(), path: path.value(i).to_string(), created: std::time::Instant::now() };
                    let _ = sender.send(cmd).await;
                }
            }
        }
        MetricBuilder::new(&self.metrics).output_rows(0).add(out.iter().map(|p| p.num_rows as usize).sum());
        Ok(out)
    }

    fn collect_plan_metrics(&self) -> Vec<MetricsSet> {
        // one MetricsSet per node below the writer, in the order utils::collect_plan_metrics walks them (utils.rs:470-481)
        let mut buf = vec![0u8; 1 << 16];
        let plan = self.plan.lock().unwrap();
        if unsafe { gpuq_plan_metrics(plan.0, buf.as_mut_ptr() as *mut _, buf.len()) } != GPUQ_OK { return vec![]; }
        let text = String::from_utf8_lossy(&buf[..buf.iter().position(|b| *b == 0).unwrap_or(0)]).into_owned();
        let nodes: Vec<serde_json::Value> = serde_json::from_str(&text).unwrap_or_default();
        nodes.iter().skip(1).map(|n| {
            let set = ExecutionPlanMetricsSet::new();
            let rows: Count = MetricBuilder::new(&set).output_rows(0);
            rows.add(n["output_rows"].as_u64().unwrap_or(0) as usize);
            let t: Time = MetricBuilder::new(&set).elapsed_compute(0);
            t.add_duration(std::time::Duration::from_nanos(n["elapsed_compute"].as_u64().unwrap_or(0)));
            set.clone_inner()
        }).collect()
    }

    fn schema(&self) -> SchemaRef {
        // ShuffleWriterExec::schema() is its child's (shuffle_writer.rs:481-483); known before execution
        // (gpuq_plan_schema types the same tree on the host and is checked against this one in tests/)
        self.stage_schema.clone()
    }
}

fn export_result(ctx: *mut gpuq_ctx, res: *mut gpuq_result) -> Result<RecordBatch> {
    let nc = unsafe { gpuq_result_num_columns(res) };
    let mut cols = vec![unsafe { std::mem::zeroed::<gpuq_column>() }; nc as usize];
    let mut fields = vec![unsafe { std::mem::zeroed::<gpuq_field_info>() }; nc as usize];
    for i in 0..nc { unsafe { gpuq_result_column(res, i, &mut cols[i as usize], &mut fields[i as usize]) }; }
    let mut arr = FFI_ArrowArray::empty();
    let mut sch = FFI_ArrowSchema::empty();
    let rc = unsafe { gpuq_export_arrow(ctx, std::ptr::null_mut(), cols.as_ptr(), fields.as_ptr(), nc, gpuq_result_num_rows(res), &mut arr, &mut sch) };
    if rc != GPUQ_OK { return Err(DataFusionError::Execution("gpuq_export_arrow failed".to_string())); }
    let data = unsafe { from_ffi(arr, &sch) }?;
    Ok(RecordBatch::from(StructArray::from(data)))
}

/// The result schema the native ShuffleWriterExec returns (flat; the reference nests the three counters in a struct,
/// shuffle_writer.rs:590-597 -- `execute_shuffle_write`, the path the engine boundary uses, returns the struct list built above).
pub fn result_schema() -> SchemaRef {
    Arc::new(Schema::new(vec![
        Field::new("partition", DataType::UInt32, false),
        Field::new("path", DataType::Utf8, false),
        Field::new("num_rows", DataType::UInt64, false),
        Field::new("num_batches", DataType::UInt64, false),
        Field::new("num_bytes", DataType::UInt64, false),
    ]))
}
