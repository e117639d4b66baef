"""Benchmark support (the counterpart of the reference's benchmarks/ directory: benchmarks/src/bin/tpch.rs, benchmarks/queries/):
TPC-H-shaped synthetic tables (SURVEY.md section 8d generator) and the physical plans of the harness's queries.  Used by bench.py,
bench_extras.py and the test-suite; nothing under arrow-ballista_amd/ imports it."""
