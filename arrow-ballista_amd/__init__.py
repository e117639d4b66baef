"""gpuq -- MI355X-native columnar operator engine behind Ballista's ExecutionEngine boundary.

This package is the Python driver around ``libgpuq.so`` (C ABI: ``include/gpuq.h``).  PyTorch is
used only as plumbing (device memory, streams, torch.distributed); every operator runs in the
hand-written HIP kernels of the shared library.  There is no CPU fallback: importing works
anywhere (so descriptors can be compiled and validated on a host without a GPU), but creating a
:class:`Context` without a HIP device raises.
"""
from .binding import (  # noqa: F401
    GpuqError, Context, Op, JoinTable, lib, lib_path, compile_check, compile_jit_source, memory_limit, memory_stats,
    T_NULL, T_BOOL, T_INT32, T_INT64, T_DATE32, T_FLOAT64, T_DECIMAL128, T_UTF8, T_UINT32, T_UINT64,
)
from . import expr  # noqa: F401
from .native import NativePlan, NativeResult  # noqa: F401
from .table import DeviceColumn, DeviceTable  # noqa: F401
from .plan import (  # noqa: F401
    MemoryExec, FilterExec, ProjectionExec, AggregateExec, HashJoinExec, CrossJoinExec, SortExec, CoalesceBatchesExec,
    RepartitionExec, ShuffleWriterExec, ShuffleReaderExec, DefaultExecutionEngine, TaskContext,
    CoalesceTasksExec, CoalescePartitionsExec, SortPreservingMergeExec, UnionExec, LocalLimitExec, GlobalLimitExec,
    RepartitionExchangeExec, BroadcastExec, RangeRepartitionExec,
)

__all__ = [
    "GpuqError", "Context", "Op", "JoinTable", "lib", "lib_path", "compile_check", "memory_limit", "memory_stats", "expr",
    "DeviceColumn", "DeviceTable", "MemoryExec", "FilterExec", "ProjectionExec", "AggregateExec",
    "HashJoinExec", "CrossJoinExec", "SortExec", "CoalesceBatchesExec", "RepartitionExec", "ShuffleWriterExec", "ShuffleReaderExec",
    "DefaultExecutionEngine", "TaskContext", "CoalesceTasksExec", "CoalescePartitionsExec", "SortPreservingMergeExec",
    "UnionExec", "LocalLimitExec", "GlobalLimitExec", "RepartitionExchangeExec", "BroadcastExec", "RangeRepartitionExec",
]
