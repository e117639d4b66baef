//! DataFusion physical plan -> the JSON mirror of `PhysicalPlanNode` that `gpuq_plan_create` executes
//! (grammar: include/gpuq.h "native plan executor"; node / field names follow ballista/core/proto/datafusion.proto).
//!
//! Nodes are recognised the way the reference does it itself: `as_any().downcast_ref::<T>()`
//! (ballista/core/src/physical_optimizer/task_group.rs:143-167, ballista/core/src/utils.rs:274-311).  A subtree the
//! device path does not execute (a scan, a window, an expression outside the supported subset) is NOT an error: it
//! becomes a *host leaf* -- the shim runs that subtree with DataFusion as the stock executor would and hands its batches
//! to the device through the Arrow C Data Interface (`gpuq_table_import_arrow`), as input slot k of a `MemoryExec` node.
use std::sync::Arc;

use ballista_core::execution_plans::{CoalesceTasksExec, ShuffleReaderExec, ShuffleWriterExec};
use datafusion::arrow::datatypes::{DataType, SchemaRef};
use datafusion::common::{DataFusionError, Result, ScalarValue};
use datafusion::logical_expr::Operator;
use datafusion::physical_plan::aggregates::{AggregateExec, AggregateMode};
use datafusion::physical_plan::coalesce_batches::CoalesceBatchesExec;
use datafusion::physical_plan::coalesce_partitions::CoalescePartitionsExec;
use datafusion::physical_plan::expressions::{
    Avg, BinaryExpr, CaseExpr, CastExpr, Column, Count, InListExpr, IsNotNullExpr, IsNullExpr, LikeExpr, Literal, Max, Min, NegativeExpr, NotExpr,
    PhysicalSortExpr, Sum, TryCastExpr,
};
use datafusion::physical_plan::filter::FilterExec;
use datafusion::physical_plan::joins::{CrossJoinExec, HashJoinExec, PartitionMode};
use datafusion::physical_plan::limit::{GlobalLimitExec, LocalLimitExec};
use datafusion::physical_plan::projection::ProjectionExec;
use datafusion::physical_plan::sorts::sort::SortExec;
use datafusion::physical_plan::sorts::sort_preserving_merge::SortPreservingMergeExec;
use datafusion::physical_plan::union::UnionExec;
use datafusion::physical_plan::{AggregateExpr, ExecutionPlan, Partitioning, PhysicalExpr};
use serde_json::{json, Value};

/// What `walk` produces: the plan JSON plus the subtrees the host must run and feed in as input slots.
pub struct WalkedPlan {
    pub json: Value,
    /// host_leaves[k] = (subtree, partition of it) feeds input slot k
    pub host_leaves: Vec<(Arc<dyn ExecutionPlan>, usize)>,
}

pub fn walk_stage(writer: &ShuffleWriterExec, job_id: &str, stage_id: usize, work_dir: &str) -> Result<WalkedPlan> {
    let mut leaves = vec![];
    let input = walk(&writer.children()[0], &mut leaves)?;
    let mut node = json!({"input": input, "job_id": job_id, "stage_id": stage_id, "work_dir": work_dir,
                          "partitions": writer.partitions()});
    if let Some(Partitioning::Hash(exprs, n)) = writer.shuffle_output_partitioning() {
        let he: Result<Vec<Value>> = exprs.iter().map(expr).collect();
        node["output_partitioning"] = json!({"hash_expr": he?, "partition_count": n});
    }
    Ok(WalkedPlan { json: json!({"ShuffleWriterExec": node}), host_leaves: leaves })
}

fn unsupported<T>(what: impl Into<String>) -> Result<T> {
    Err(DataFusionError::NotImplemented(what.into()))
}

pub fn type_json(t: &DataType) -> Result<Value> {
    Ok(match t {
        DataType::Boolean => json!("Boolean"),
        DataType::Int32 => json!("Int32"),
        DataType::Int64 => json!("Int64"),
        DataType::UInt32 => json!("UInt32"),
        DataType::UInt64 => json!("UInt64"),
        DataType::Date32 => json!("Date32"),
        DataType::Float64 => json!("Float64"),
        DataType::Utf8 => json!("Utf8"),
        DataType::Decimal128(p, s) => json!({"Decimal128": [p, s]}),
        DataType::Int8 => json!("Int8"),
        DataType::Int16 => json!("Int16"),
        DataType::UInt8 => json!("UInt8"),
        DataType::UInt16 => json!("UInt16"),
        DataType::Float32 => json!("Float32"),
        DataType::Date64 => json!("Date64"),
        // the engine keeps the unit; the zone travels in the descriptor and comes back with this plan's schema (batches are built with it)
        DataType::Timestamp(u, tz) => json!({"Timestamp": [format!("{u:?}"), tz.as_ref().map(|z| z.to_string())]}),
        // converted where the batches enter (gpuq_table_import_arrow / gpuq_ingest_push): inside, 32-bit offsets / the value type
        DataType::LargeUtf8 => json!("LargeUtf8"),
        DataType::Dictionary(k, v) => json!({"Dictionary": [type_json(k)?, type_json(v)?]}),
        other => return unsupported(format!("column type {other:?}")),
    })
}

pub fn schema_json(s: &SchemaRef) -> Result<Value> {
    let f: Result<Vec<Value>> = s.fields().iter().map(|f| Ok(json!({"name": f.name(), "type": type_json(f.data_type())?, "nullable": f.is_nullable()}))).collect();
    Ok(Value::Array(f?))
}

fn literal(v: &ScalarValue) -> Result<Value> {
    // decimals travel as the unscaled integer in a string; NULL literals keep their type
    Ok(match v {
        ScalarValue::Boolean(x) => json!({"type": "Boolean", "value": x}),
        ScalarValue::Int32(x) => json!({"type": "Int32", "value": x}),
        ScalarValue::Int64(x) => json!({"type": "Int64", "value": x}),
        ScalarValue::UInt32(x) => json!({"type": "UInt32", "value": x}),
        ScalarValue::UInt64(x) => json!({"type": "UInt64", "value": x}),
        ScalarValue::Date32(x) => json!({"type": "Date32", "value": x}),
        ScalarValue::Float64(x) => json!({"type": "Float64", "value": x}),
        ScalarValue::Utf8(x) => json!({"type": "Utf8", "value": x}),
        ScalarValue::LargeUtf8(x) => json!({"type": "Utf8", "value": x}),
        ScalarValue::Int8(x) => json!({"type": "Int8", "value": x}),
        ScalarValue::Int16(x) => json!({"type": "Int16", "value": x}),
        ScalarValue::UInt8(x) => json!({"type": "UInt8", "value": x}),
        ScalarValue::UInt16(x) => json!({"type": "UInt16", "value": x}),
        ScalarValue::Float32(x) => json!({"type": "Float32", "value": x}),
        ScalarValue::Date64(x) => json!({"type": "Date64", "value": x.map(|v| v.to_string())}),
        ScalarValue::TimestampSecond(x, tz) => json!({"type": {"Timestamp": ["Second", tz.as_ref().map(|z| z.to_string())]}, "value": x.map(|v| v.to_string())}),
        ScalarValue::TimestampMillisecond(x, tz) => json!({"type": {"Timestamp": ["Millisecond", tz.as_ref().map(|z| z.to_string())]}, "value": x.map(|v| v.to_string())}),
        ScalarValue::TimestampMicrosecond(x, tz) => json!({"type": {"Timestamp": ["Microsecond", tz.as_ref().map(|z| z.to_string())]}, "value": x.map(|v| v.to_string())}),
        ScalarValue::TimestampNanosecond(x, tz) => json!({"type": {"Timestamp": ["Nanosecond", tz.as_ref().map(|z| z.to_string())]}, "value": x.map(|v| v.to_string())}),
        ScalarValue::Decimal128(x, p, s) => json!({"type": {"Decimal128": [p, s]}, "value": x.map(|v| v.to_string())}),
        other => return unsupported(format!("literal {other:?}")),
    })
}

fn binary_op(op: &Operator) -> Result<&'static str> {
    // PhysicalBinaryExprNode.op travels as the operator's name (datafusion.proto:1228-1232)
    Ok(match op {
        Operator::Plus => "Plus", Operator::Minus => "Minus", Operator::Multiply => "Multiply", Operator::Divide => "Divide", Operator::Modulo => "Modulo",
        Operator::Eq => "Eq", Operator::NotEq => "NotEq", Operator::Lt => "Lt", Operator::LtEq => "LtEq", Operator::Gt => "Gt", Operator::GtEq => "GtEq",
        Operator::And => "And", Operator::Or => "Or",
        other => return unsupported(format!("binary operator {other:?}")),
    })
}

/// PhysicalExprNode mirror (datafusion.proto:1142-1180).
pub fn expr(e: &Arc<dyn PhysicalExpr>) -> Result<Value> {
    let a = e.as_any();
    if let Some(c) = a.downcast_ref::<Column>() {
        return Ok(json!({"column": {"name": c.name(), "index": c.index()}}));
    }
    if let Some(l) = a.downcast_ref::<Literal>() {
        return Ok(json!({"literal": literal(l.value())?}));
    }
    if let Some(b) = a.downcast_ref::<BinaryExpr>() {
        return Ok(json!({"binary_expr": {"l": expr(b.left())?, "r": expr(b.right())?, "op": binary_op(b.op())?}}));
    }
    if let Some(c) = a.downcast_ref::<CastExpr>() {
        return Ok(json!({"cast": {"expr": expr(c.expr())?, "arrow_type": type_json(c.cast_type())?}}));
    }
    if let Some(c) = a.downcast_ref::<TryCastExpr>() {
        return Ok(json!({"try_cast": {"expr": expr(c.expr())?, "arrow_type": type_json(c.cast_type())?}}));
    }
    if let Some(x) = a.downcast_ref::<IsNullExpr>() {
        return Ok(json!({"is_null_expr": {"expr": expr(x.arg())?}}));
    }
    if let Some(x) = a.downcast_ref::<IsNotNullExpr>() {
        return Ok(json!({"is_not_null_expr": {"expr": expr(x.arg())?}}));
    }
    if let Some(x) = a.downcast_ref::<NotExpr>() {
        return Ok(json!({"not_expr": {"expr": expr(x.arg())?}}));
    }
    if let Some(x) = a.downcast_ref::<NegativeExpr>() {
        return Ok(json!({"negative": {"expr": expr(x.arg())?}}));
    }
    if let Some(x) = a.downcast_ref::<InListExpr>() {
        let list: Result<Vec<Value>> = x.list().iter().map(expr).collect();
        return Ok(json!({"in_list": {"expr": expr(x.expr())?, "list": list?, "negated": x.negated()}}));
    }
    if let Some(x) = a.downcast_ref::<CaseExpr>() {
        let wt: Result<Vec<Value>> = x.when_then_expr().iter().map(|(w, t)| Ok(json!({"when_expr": expr(w)?, "then_expr": expr(t)?}))).collect();
        let base = match x.expr() { Some(b) => expr(b)?, None => Value::Null };
        let els = match x.else_expr() { Some(b) => expr(b)?, None => Value::Null };
        return Ok(json!({"case_": {"expr": base, "when_then_expr": wt?, "else_expr": els}}));
    }
    if let Some(x) = a.downcast_ref::<LikeExpr>() {
        return Ok(json!({"like_expr": {"negated": x.negated(), "case_insensitive": x.case_insensitive(), "expr": expr(x.expr())?, "pattern": expr(x.pattern())?}}));
    }
    unsupported(format!("physical expression {e:?}"))
}

fn sort_exprs(v: &[PhysicalSortExpr]) -> Result<Value> {
    let r: Result<Vec<Value>> = v.iter().map(|s| Ok(json!({"expr": expr(&s.expr)?, "asc": !s.options.descending, "nulls_first": s.options.nulls_first}))).collect();
    Ok(Value::Array(r?))
}

fn aggregate_fn(a: &Arc<dyn AggregateExpr>) -> Result<&'static str> {
    let x = a.as_any();
    Ok(if x.is::<Sum>() { "SUM" } else if x.is::<Avg>() { "AVG" } else if x.is::<Count>() { "COUNT" } else if x.is::<Min>() { "MIN" } else if x.is::<Max>() { "MAX" }
       else { return unsupported(format!("aggregate {}", a.name())) })
}

/// One plan node.  On `NotImplemented` from anything below, the caller turns the WHOLE subtree rooted here into a host leaf.
fn walk_node(p: &Arc<dyn ExecutionPlan>, leaves: &mut Vec<(Arc<dyn ExecutionPlan>, usize)>) -> Result<Value> {
    let a = p.as_any();
    if let Some(n) = a.downcast_ref::<CoalesceBatchesExec>() {
        return Ok(json!({"CoalesceBatchesExec": {"input": walk(n.input(), leaves)?}}));
    }
    if let Some(n) = a.downcast_ref::<FilterExec>() {
        return Ok(json!({"FilterExec": {"input": walk(n.input(), leaves)?, "expr": expr(n.predicate())?}}));
    }
    if let Some(n) = a.downcast_ref::<ProjectionExec>() {
        let e: Result<Vec<Value>> = n.expr().iter().map(|(e, _)| expr(e)).collect();
        let names: Vec<&str> = n.expr().iter().map(|(_, s)| s.as_str()).collect();
        return Ok(json!({"ProjectionExec": {"input": walk(n.input(), leaves)?, "expr": e?, "expr_name": names}}));
    }
    if let Some(n) = a.downcast_ref::<AggregateExec>() {
        if n.group_expr().null_expr().iter().any(|_| true) && n.group_expr().groups().len() > 1 {
            return unsupported("grouping sets");
        }
        let mode = match n.mode() {
            AggregateMode::Partial => "Partial", AggregateMode::Final => "Final", AggregateMode::FinalPartitioned => "FinalPartitioned",
            AggregateMode::Single => "Single", AggregateMode::SinglePartitioned => "Single",
        };
        let final_ = matches!(n.mode(), AggregateMode::Final | AggregateMode::FinalPartitioned);
        let ge: Result<Vec<Value>> = n.group_expr().expr().iter().map(|(e, name)| Ok(json!({"expr": expr(e)?, "name": name}))).collect();
        let mut ae = vec![];
        for agg in n.aggr_expr() {
            let mut o = json!({"fn": aggregate_fn(agg)?, "name": agg.name()});
            if !final_ {
                // Final modes read the state columns of the Partial stage by position: no argument expressions
                let args = agg.expressions();
                if let Some(e0) = args.first() { o["expr"] = expr(e0)?; }
                if let Some(e1) = args.get(1) { o["expr2"] = expr(e1)?; }
            }
            ae.push(o);
        }
        if n.filter_expr().iter().any(|f| f.is_some()) { return unsupported("aggregate FILTER clause"); }
        return Ok(json!({"AggregateExec": {"input": walk(n.input(), leaves)?, "mode": mode, "group_expr": ge?, "aggr_expr": ae}}));
    }
    if let Some(n) = a.downcast_ref::<HashJoinExec>() {
        let on: Vec<Value> = n.on().iter().map(|(l, r)| json!({"left": {"column": {"name": l.name(), "index": l.index()}},
                                                                "right": {"column": {"name": r.name(), "index": r.index()}}})).collect();
        let mode = match n.partition_mode() { PartitionMode::CollectLeft => "CollectLeft", PartitionMode::Partitioned => "Partitioned", PartitionMode::Auto => "CollectLeft" };
        let mut j = json!({"left": walk(n.left(), leaves)?, "right": walk(n.right(), leaves)?, "on": on, "join_type": format!("{:?}", n.join_type()),
                           "partition_mode": mode, "null_equals_null": n.null_equals_null()});
        if let Some(f) = n.filter() {
            // JoinFilter.expression is written against the intermediate schema (column_indices); the device evaluates it over
            // the joined row, so columns are re-pointed at (side, index) -> position in left ++ right
            j["filter"] = crate::plan_walk::join_filter(f, n.left().schema().fields().len())?;
        }
        return Ok(json!({"HashJoinExec": j}));
    }
    if let Some(n) = a.downcast_ref::<CrossJoinExec>() {
        // (an uncorrelated scalar subquery arrives as a one-row left side; gpuq refuses more than 2^32 pairs)
        return Ok(json!({"CrossJoinExec": {"left": walk(n.left(), leaves)?, "right": walk(n.right(), leaves)?}}));
    }
    if let Some(n) = a.downcast_ref::<SortExec>() {
        if n.preserve_partitioning() && n.input().output_partitioning().partition_count() > 1 {
            // per-partition sort: one task covers the partitions CoalesceTasksExec hands it; the native SortExec sorts what it is given
        }
        return Ok(json!({"SortExec": {"input": walk(n.input(), leaves)?, "expr": sort_exprs(n.expr())?, "fetch": n.fetch().map(|f| f as i64).unwrap_or(-1)}}));
    }
    if let Some(n) = a.downcast_ref::<SortPreservingMergeExec>() {
        return Ok(json!({"SortPreservingMergeExec": {"input": walk(n.input(), leaves)?, "expr": sort_exprs(n.expr())?, "fetch": n.fetch().map(|f| f as i64).unwrap_or(-1)}}));
    }
    if let Some(n) = a.downcast_ref::<CoalescePartitionsExec>() {
        return Ok(json!({"CoalescePartitionsExec": {"input": walk(n.input(), leaves)?}}));
    }
    if let Some(n) = a.downcast_ref::<UnionExec>() {
        let ins: Result<Vec<Value>> = n.inputs().iter().map(|i| walk(i, leaves)).collect();
        return Ok(json!({"UnionExec": {"inputs": ins?}}));
    }
    if let Some(n) = a.downcast_ref::<LocalLimitExec>() {
        return Ok(json!({"LocalLimitExec": {"input": walk(n.input(), leaves)?, "fetch": n.fetch()}}));
    }
    if let Some(n) = a.downcast_ref::<GlobalLimitExec>() {
        return Ok(json!({"GlobalLimitExec": {"input": walk(n.input(), leaves)?, "skip": n.skip(), "fetch": n.fetch().map(|f| f as i64).unwrap_or(-1)}}));
    }
    if let Some(n) = a.downcast_ref::<CoalesceTasksExec>() {
        let mut j = json!({"input": walk(&n.children()[0], leaves)?, "partitions": n.partitions()});
        if let Some(ob) = n.order_by() { j["order_by"] = sort_exprs(ob)?; }
        return Ok(json!({"CoalesceTasksExec": j}));
    }
    if let Some(n) = a.downcast_ref::<ShuffleReaderExec>() {
        // local files are decoded on the device; a location that is not on this node's filesystem ("local" is decided exactly
        // as the reference does, Path::exists, shuffle_reader.rs:626-628) makes the reader a host leaf: DataFusion's
        // ShuffleReaderExec fetches it over Flight and the batches are imported
        let mut parts = vec![];
        for locs in &n.partition {
            let mut files = vec![];
            for l in locs {
                if l.partition_stats.num_rows() == Some(0) { continue; }                 // shuffle_reader.rs:251
                if !std::path::Path::new(&l.path).exists() { return unsupported("remote shuffle location"); }
                files.push(json!({"path": l.path}));
            }
            parts.push(Value::Array(files));
        }
        return Ok(json!({"ShuffleReaderExec": {"schema": schema_json(&n.schema())?, "partition": parts}}));
    }
    unsupported(format!("plan node {}", p.name()))
}

pub fn walk(p: &Arc<dyn ExecutionPlan>, leaves: &mut Vec<(Arc<dyn ExecutionPlan>, usize)>) -> Result<Value> {
    let mark = leaves.len();
    match walk_node(p, leaves) {
        Ok(v) => Ok(v),
        Err(DataFusionError::NotImplemented(why)) => {
            // this subtree stays on the host: forget host leaves collected below it, it becomes one leaf itself
            leaves.truncate(mark);
            log::debug!("gpuq: subtree rooted at {} runs on the host ({why})", p.name());
            let nparts = p.output_partitioning().partition_count();
            let mut slots = vec![];
            for k in 0..nparts { slots.push(leaves.len()); leaves.push((p.clone(), k)); }      // input slot = position in host_leaves
            Ok(json!({"MemoryExec": {"schema": schema_json(&p.schema())?, "partitions": slots}}))
        }
        Err(e) => Err(e),
    }
}

/// JoinFilter -> expression over the joined row (left columns, then right columns).
pub fn join_filter(f: &datafusion::physical_plan::joins::utils::JoinFilter, n_left: usize) -> Result<Value> {
    use datafusion::common::JoinSide;
    let v = expr(f.expression())?;
    let idx = f.column_indices();
    fn remap(v: &mut Value, idx: &[datafusion::physical_plan::joins::utils::ColumnIndex], n_left: usize) {
        match v {
            Value::Object(m) => {
                if let Some(Value::Object(c)) = m.get_mut("column") {
                    if let Some(i) = c.get("index").and_then(|i| i.as_u64()) {
                        let ci = &idx[i as usize];
                        let pos = if ci.side == JoinSide::Left { ci.index } else { n_left + ci.index };
                        c.insert("index".into(), json!(pos));      // the native executor resolves by NAME: a filter over two sides that
                                                                     // share a column name is refused there (GPUQ_ERR_INVALID), not mis-bound
                    }
                    return;
                }
                for (_, x) in m.iter_mut() { remap(x, idx, n_left); }
            }
            Value::Array(a) => for x in a.iter_mut() { remap(x, idx, n_left); },
            _ => {}
        }
    }
    let mut v = v;
    remap(&mut v, idx, n_left);
    Ok(v)
}
