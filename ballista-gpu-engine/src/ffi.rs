//! include/gpuq.h, the part the shim calls.  Plain C ABI: pointers, sizes, POD structs.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const GPUQ_OK: c_int = 0;
pub const GPUQ_ERR_UNSUPPORTED: c_int = 3;
pub const GPUQ_ERR_CANCELLED: c_int = 6;

#[repr(C)]
pub struct gpuq_ctx { _p: [u8; 0] }
#[repr(C)]
pub struct gpuq_plan { _p: [u8; 0] }
#[repr(C)]
pub struct gpuq_result { _p: [u8; 0] }
#[repr(C)]
pub struct gpuq_task { _p: [u8; 0] }
#[repr(C)]
pub struct gpuq_table { _p: [u8; 0] }

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gpuq_column {
    pub type_: i32,
    pub precision: i32,
    pub scale: i32,
    pub repr: i32,
    pub data: *const c_void,
    pub offsets: *const i32,
    pub validity: *const u8,
    pub length: i64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gpuq_field_info {
    pub name: [c_char; 256],
    pub type_: i32,
    pub precision: i32,
    pub scale: i32,
    pub nullable: i32,
    pub repr: i32,
    pub width: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct gpuq_input {
    pub cols: *const gpuq_column,
    pub n_cols: i32,
    pub n_via: i32,
    pub n_rows: i64,
    pub via: [*const u32; 3],
    pub n_rows_dev: *const u64, // NULL: n_rows is exact (include/gpuq.h "Deferred execution")
}

extern "C" {
    pub fn gpuq_ctx_create(device_ordinal: c_int, json_opts: *const c_char) -> *mut gpuq_ctx;
    pub fn gpuq_ctx_free(ctx: *mut gpuq_ctx);
    pub fn gpuq_jit_quiesce();
    pub fn gpuq_memory_limit(bytes: i64) -> c_int;
    pub fn gpuq_memory_stats(in_use: *mut i64, peak: *mut i64, cached: *mut i64, limit: *mut i64, reset_peak: c_int) -> c_int;
    pub fn gpuq_last_error(ctx: *mut gpuq_ctx) -> *const c_char;

    pub fn gpuq_plan_create(ctx: *mut gpuq_ctx, plan_json: *const c_char, out: *mut *mut gpuq_plan) -> c_int;
    pub fn gpuq_plan_free(plan: *mut gpuq_plan);
    pub fn gpuq_plan_schema(plan: *mut gpuq_plan, fields_out: *mut gpuq_field_info, cap: c_int, n_out: *mut c_int) -> c_int;
    pub fn gpuq_plan_execute_async(plan: *mut gpuq_plan, stream: *mut c_void, partition: c_int, inputs: *const gpuq_input, n_inputs: c_int,
                                   out: *mut *mut gpuq_task) -> c_int;
    pub fn gpuq_task_poll(task: *mut gpuq_task, done_out: *mut c_int) -> c_int;
    pub fn gpuq_task_cancel(task: *mut gpuq_task) -> c_int;
    pub fn gpuq_task_wait(task: *mut gpuq_task, out: *mut *mut gpuq_result) -> c_int;
    pub fn gpuq_task_free(task: *mut gpuq_task);
    pub fn gpuq_plan_metrics(plan: *mut gpuq_plan, json_out: *mut c_char, cap: usize) -> c_int;
    pub fn gpuq_plan_last_error() -> *const c_char;

    pub fn gpuq_result_num_rows(r: *const gpuq_result) -> i64;
    pub fn gpuq_result_num_columns(r: *const gpuq_result) -> c_int;
    pub fn gpuq_result_column(r: *const gpuq_result, i: c_int, col_out: *mut gpuq_column, field_out: *mut gpuq_field_info) -> c_int;
    pub fn gpuq_result_free(r: *mut gpuq_result);

    /// Arrow C Data Interface: a RecordBatch in (copied to HBM through pinned staging), device columns out.
    pub fn gpuq_table_import_arrow(ctx: *mut gpuq_ctx, stream: *mut c_void, batch: *const arrow::ffi::FFI_ArrowArray,
                                   schema: *const arrow::ffi::FFI_ArrowSchema, out: *mut *mut gpuq_table) -> c_int;
    pub fn gpuq_table_num_rows(t: *const gpuq_table) -> i64;
    pub fn gpuq_table_num_columns(t: *const gpuq_table) -> c_int;
    pub fn gpuq_table_column(t: *const gpuq_table, i: c_int, col_out: *mut gpuq_column, field_out: *mut gpuq_field_info) -> c_int;
    pub fn gpuq_table_free(t: *mut gpuq_table);
    pub fn gpuq_export_arrow(ctx: *mut gpuq_ctx, stream: *mut c_void, cols: *const gpuq_column, fields: *const gpuq_field_info, n_cols: c_int, n_rows: i64,
                             out: *mut arrow::ffi::FFI_ArrowArray, out_schema: *mut arrow::ffi::FFI_ArrowSchema) -> c_int;
}

/// The library's per-thread error text for plan calls.
pub fn plan_error() -> String {
    unsafe {
        let p = gpuq_plan_last_error();
        if p.is_null() { "gpuq error".to_string() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}
