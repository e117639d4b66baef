/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Plain-C restatement used as the CPU checker (tests/) and as
 * bench.py's cpu_baseline ("port").  Nothing under arrow-ballista_amd/ links, loads or calls it.
 *
 * The reference's arithmetic for this path lives in third-party crates absent from the container
 * (datafusion git tag v34.0.0-cx.1, arrow 49.0.0; reference Cargo.toml:33-43), so the
 * functions below restate the published operator semantics for the TPC-H shapes the benchmark
 * harness runs (reference benchmarks/queries/q1.sql, q3.sql, q5.sql; schema
 * benchmarks/src/bin/tpch.rs:864-957) in the way a vectorised CPU engine executes them: one pass
 * over Arrow-layout columns, int128 accumulators, a chained hash table for joins (hash -> chain of
 * build rows, key re-verification, as DataFusion's HashJoinExec does [UPSTREAM-KNOWLEDGE]).
 * PARITY PINNING: see oracle/oracle_np.py header -- "parity unpinned" for everything except the
 * ungrouped aggregate KATs; this file is validated against oracle_np.py in tests/test_oracle_c.py.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef __int128 i128;
typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef int64_t i64;

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------------------------------------------------------- synthetic data */
static inline u64 mix64(u64 x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31; return x;
}
static inline u64 gen_u64(u64 seed, u64 col, u64 row) {
  return mix64((seed + col * 0xD1B54A32D192ED03ull) ^ (row * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull));
}
static inline uint32_t gen_mod(u64 seed, u64 col, u64 row, uint32_t m) {
  const u64 x = gen_u64(seed, col, row);
  return ((uint32_t)(x >> 32) ^ (uint32_t)x) % m;
}
static inline i64 order_key(i64 o) { return (o >> 3) * 32 + (o & 7) + 1; }
static inline int32_t order_date(u64 seed_orders, i64 o) { return 8035 + (int32_t)gen_mod(seed_orders, 4, (u64)o, 2406u); }

/* Decimal128 columns are written as 16-byte little-endian values (lo, hi). */
void oracle_gen_lineitem(u64 seed, u64 seed_orders, i64 row0, i64 n, i64 n_supp, i64* l_orderkey, i64* l_suppkey, u64* l_quantity,
                         u64* l_extendedprice, u64* l_discount, u64* l_tax, int32_t* l_shipdate, uint8_t* l_returnflag,
                         int32_t* l_returnflag_off, uint8_t* l_linestatus, int32_t* l_linestatus_off) {
#pragma omp parallel for schedule(static)
  for (i64 j = 0; j < n; ++j) {
    const i64 i = row0 + j, o = i >> 2;
    const int32_t odate = order_date(seed_orders, o);
    const int32_t ship = odate + 1 + (int32_t)gen_mod(seed, 1, (u64)i, 121u);
    const int32_t receipt = ship + 1 + (int32_t)gen_mod(seed, 2, (u64)i, 30u);
    if (l_orderkey) l_orderkey[j] = order_key(o);
    if (l_suppkey) l_suppkey[j] = 1 + (i64)gen_mod(seed, 3, (u64)i, (uint32_t)n_supp);
    if (l_quantity) { l_quantity[2 * j] = (u64)((gen_mod(seed, 4, (u64)i, 50u) + 1u) * 100u); l_quantity[2 * j + 1] = 0; }
    if (l_extendedprice) { l_extendedprice[2 * j] = (u64)(90100u + gen_mod(seed, 5, (u64)i, 10404851u)); l_extendedprice[2 * j + 1] = 0; }
    if (l_discount) { l_discount[2 * j] = (u64)gen_mod(seed, 6, (u64)i, 11u); l_discount[2 * j + 1] = 0; }
    if (l_tax) { l_tax[2 * j] = (u64)gen_mod(seed, 7, (u64)i, 9u); l_tax[2 * j + 1] = 0; }
    if (l_shipdate) l_shipdate[j] = ship;
    if (l_returnflag) { l_returnflag[j] = (receipt <= 9298) ? ((gen_u64(seed, 8, (u64)i) & 1) ? 'R' : 'A') : 'N'; l_returnflag_off[j] = (int32_t)j; }
    if (l_linestatus) { l_linestatus[j] = (ship > 9298) ? 'O' : 'F'; l_linestatus_off[j] = (int32_t)j; }
  }
  if (l_returnflag) l_returnflag_off[n] = (int32_t)n;
  if (l_linestatus) l_linestatus_off[n] = (int32_t)n;
}

void oracle_gen_orders(u64 seed, i64 row0, i64 n, i64 n_cust, i64* o_orderkey, i64* o_custkey, int32_t* o_orderdate, int32_t* o_shippriority) {
#pragma omp parallel for schedule(static)
  for (i64 j = 0; j < n; ++j) {
    const i64 o = row0 + j;
    if (o_orderkey) o_orderkey[j] = order_key(o);
    if (o_custkey) o_custkey[j] = 3 * (i64)gen_mod(seed, 2, (u64)o, (uint32_t)(n_cust / 3)) + 1 + (i64)((gen_u64(seed, 2, (u64)o) >> 40) & 1);
    if (o_orderdate) o_orderdate[j] = order_date(seed, o);
    if (o_shippriority) o_shippriority[j] = 0;
  }
}

static const char* kSegments[5] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"};
static const int kSegLen[5] = {10, 8, 9, 9, 9};
static void seg_perm(u64 x64, int perm[5]) {
  uint32_t x = (uint32_t)(x64 >> 32) ^ (uint32_t)x64;
  int pool[5] = {0, 1, 2, 3, 4};
  for (int k = 0; k < 5; ++k) {
    const int r = (int)(x % (uint32_t)(5 - k)); x /= (uint32_t)(5 - k);
    int pick = 0, seen = 0;
    for (int q = 0; q < 5; ++q) if (pool[q] >= 0) { if (seen == r) pick = q; ++seen; }
    perm[k] = pool[pick]; pool[pick] = -1;
  }
}
/* row0 multiple of 5; c_mktsegment needs 45 bytes per 5 rows */
void oracle_gen_customer(u64 seed, i64 row0, i64 n, i64* c_custkey, i64* c_nationkey, uint8_t* c_mktsegment, int32_t* c_mktsegment_off) {
  for (i64 j = 0; j < n; ++j) {
    const i64 i = row0 + j;
    if (c_custkey) c_custkey[j] = i + 1;
    if (c_nationkey) c_nationkey[j] = (i64)gen_mod(seed, 2, (u64)i, 25u);
    if (c_mktsegment) {
      const i64 run = i / 5; const int m = (int)(i % 5);
      int perm[5]; seg_perm(gen_u64(seed, 3, (u64)run), perm);
      int off = 0; for (int k = 0; k < m; ++k) off += kSegLen[perm[k]];
      const int seg = perm[m];
      const i64 base = (run - row0 / 5) * 45 + off;
      c_mktsegment_off[j] = (int32_t)base;
      memcpy(c_mktsegment + base, kSegments[seg], (size_t)kSegLen[seg]);
      if (j == n - 1) c_mktsegment_off[n] = (int32_t)(base + kSegLen[seg]);
    }
  }
}
void oracle_gen_supplier(u64 seed, i64 row0, i64 n, i64* s_suppkey, i64* s_nationkey) {
  for (i64 j = 0; j < n; ++j) {
    const i64 i = row0 + j;
    if (s_suppkey) s_suppkey[j] = i + 1;
    if (s_nationkey) s_nationkey[j] = (i64)gen_mod(seed, 2, (u64)i, 25u);
  }
}

/* ---------------------------------------------------------------- q1: filter + project + group-by
 * reference benchmarks/queries/q1.sql over the Arrow-physical columns (Decimal128 16 B, Utf8 offsets+bytes).
 * Output: up to 8 groups; per group: key bytes (rf, ls), and 128-bit sums as (lo,hi):
 *   sum_qty(s2) sum_base(s2) sum_disc_price(s4) sum_charge(s6) sum_disc(s2) count
 * AVGs are derived by the caller: sum*10^4/count truncated (DataFusion Decimal avg). */
typedef struct { uint8_t rf, ls; i128 s_qty, s_base, s_dp, s_ch, s_disc; i64 cnt; int used; } q1_group;

int oracle_q1(i64 n, const u64* l_quantity, const u64* l_extendedprice, const u64* l_discount, const u64* l_tax, const int32_t* l_shipdate,
              const uint8_t* l_returnflag, const int32_t* rf_off, const uint8_t* l_linestatus, const int32_t* ls_off, int32_t ship_max,
              uint8_t* out_keys /*[8][2]*/, u64* out_sums /*[8][5][2]*/, i64* out_counts /*[8]*/) {
  int nthreads = oracle_num_threads();
  q1_group* all = (q1_group*)calloc((size_t)nthreads * 65536, sizeof(q1_group));   /* direct-addressed by the two key bytes */
#pragma omp parallel
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    q1_group* g = all + (size_t)tid * 65536;
#pragma omp for schedule(static)
    for (i64 i = 0; i < n; ++i) {
      if (l_shipdate[i] > ship_max) continue;
      const uint8_t rf = l_returnflag[rf_off[i]], ls = l_linestatus[ls_off[i]];
      q1_group* q = &g[(rf << 8) | ls];
      const i128 qty = (i128)(((u128)l_quantity[2 * i + 1] << 64) | l_quantity[2 * i]);
      const i128 ext = (i128)(((u128)l_extendedprice[2 * i + 1] << 64) | l_extendedprice[2 * i]);
      const i128 disc = (i128)(((u128)l_discount[2 * i + 1] << 64) | l_discount[2 * i]);
      const i128 tax = (i128)(((u128)l_tax[2 * i + 1] << 64) | l_tax[2 * i]);
      const i128 dp = ext * (100 - disc);        /* Decimal(38,4) */
      const i128 ch = dp * (100 + tax);          /* Decimal(38,6) */
      q->rf = rf; q->ls = ls; q->used = 1;
      q->s_qty += qty; q->s_base += ext; q->s_dp += dp; q->s_ch += ch; q->s_disc += disc; q->cnt += 1;
    }
  }
  int ng = 0;
  for (int k = 0; k < 65536; ++k) {
    q1_group acc; memset(&acc, 0, sizeof(acc));
    for (int t = 0; t < nthreads; ++t) {
      const q1_group* q = &all[(size_t)t * 65536 + k];
      if (!q->used) continue;
      acc.used = 1; acc.rf = q->rf; acc.ls = q->ls;
      acc.s_qty += q->s_qty; acc.s_base += q->s_base; acc.s_dp += q->s_dp; acc.s_ch += q->s_ch; acc.s_disc += q->s_disc; acc.cnt += q->cnt;
    }
    if (!acc.used) continue;
    if (ng < 8) {
      out_keys[2 * ng] = acc.rf; out_keys[2 * ng + 1] = acc.ls;
      const i128 v[5] = {acc.s_qty, acc.s_base, acc.s_dp, acc.s_ch, acc.s_disc};
      for (int a = 0; a < 5; ++a) { out_sums[(ng * 5 + a) * 2] = (u64)v[a]; out_sums[(ng * 5 + a) * 2 + 1] = (u64)((u128)v[a] >> 64); }
      out_counts[ng] = acc.cnt;
    }
    ++ng;
  }
  free(all);
  return ng;
}

/* ---------------------------------------------------------------- hash join (inner, int64 keys)
 * Build: chained table hash -> head row, next[] per build row (DataFusion JoinHashMap shape).
 * Probe: for every probe key walk the chain, verify the key, emit (build,probe) pairs.
 * Returns the pair count; pairs are written when out_build != NULL (capacity cap). */
typedef struct { u64 mask; uint32_t* head; uint32_t* next; const i64* keys; i64 n; } oracle_join_table;

void* oracle_join_build(const i64* keys, i64 n) {
  oracle_join_table* t = (oracle_join_table*)malloc(sizeof(*t));
  u64 slots = 1024; while (slots < (u64)n * 2) slots <<= 1;
  t->mask = slots - 1; t->keys = keys; t->n = n;
  t->head = (uint32_t*)malloc(slots * 4); memset(t->head, 0xFF, slots * 4);
  t->next = (uint32_t*)malloc((size_t)(n > 0 ? n : 1) * 4);
  for (i64 i = 0; i < n; ++i) {
    const u64 h = mix64((u64)keys[i]) & t->mask;
    t->next[i] = t->head[h]; t->head[h] = (uint32_t)i;
  }
  return t;
}
void oracle_join_free(void* tv) { oracle_join_table* t = (oracle_join_table*)tv; free(t->head); free(t->next); free(t); }

i64 oracle_join_probe(void* tv, const i64* probe, i64 n, uint32_t* out_build, uint32_t* out_probe, i64 cap, u64* checksum_out) {
  const oracle_join_table* t = (const oracle_join_table*)tv;
  i64 total = 0; u64 checksum = 0;
  if (out_build) {   /* ordered, single thread: exact pair list */
    for (i64 j = 0; j < n; ++j) {
      const i64 k = probe[j];
      for (uint32_t r = t->head[mix64((u64)k) & t->mask]; r != 0xFFFFFFFFu; r = t->next[r])
        if (t->keys[r] == k) { if (total < cap) { out_build[total] = r; out_probe[total] = (uint32_t)j; } ++total; checksum += mix64(((u64)r << 32) | (u64)j); }
    }
  } else {
#pragma omp parallel for schedule(static) reduction(+ : total, checksum)
    for (i64 j = 0; j < n; ++j) {
      const i64 k = probe[j];
      for (uint32_t r = t->head[mix64((u64)k) & t->mask]; r != 0xFFFFFFFFu; r = t->next[r])
        if (t->keys[r] == k) { ++total; checksum += mix64(((u64)r << 32) | (u64)j); }
    }
  }
  if (checksum_out) *checksum_out = checksum;
  return total;
}

/* ---------------------------------------------------------------- sort: stable LSD radix on (u64 key, u32 id) */
void oracle_sort_u64(const u64* keys, i64 n, uint32_t* perm_out) {
  u64* ka = (u64*)malloc((size_t)n * 8); u64* kb = (u64*)malloc((size_t)n * 8);
  uint32_t* va = (uint32_t*)malloc((size_t)n * 4); uint32_t* vb = (uint32_t*)malloc((size_t)n * 4);
  memcpy(ka, keys, (size_t)n * 8);
  for (i64 i = 0; i < n; ++i) va[i] = (uint32_t)i;
  for (int sh = 0; sh < 64; sh += 8) {
    i64 cnt[257]; memset(cnt, 0, sizeof(cnt));
    for (i64 i = 0; i < n; ++i) cnt[((ka[i] >> sh) & 0xFF) + 1]++;
    if (cnt[((ka[0] >> sh) & 0xFF) + 1] == n) continue;   /* constant digit */
    for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
    for (i64 i = 0; i < n; ++i) { const i64 p = cnt[(ka[i] >> sh) & 0xFF]++; kb[p] = ka[i]; vb[p] = va[i]; }
    u64* tk = ka; ka = kb; kb = tk; uint32_t* tv = va; va = vb; vb = tv;
  }
  memcpy(perm_out, va, (size_t)n * 4);
  free(ka); free(kb); free(va); free(vb);
}

/* ---------------------------------------------------------------- hash partition ids (gpuq's mix64 hash, int64 key) */
static inline u64 hash_combine1(u64 h, u64 lo) {
  const u64 v = mix64(lo ^ mix64(0x632BE59BD9B4E019ull));
  return mix64(h * 31 + v + 0x9E3779B97F4A7C15ull);
}
void oracle_partition_ids_i64(const i64* keys, i64 n, uint32_t nparts, uint32_t* pid_out) {
#pragma omp parallel for schedule(static)
  for (i64 i = 0; i < n; ++i) pid_out[i] = (uint32_t)(hash_combine1(0x243F6A8885A308D3ull, (u64)keys[i]) % nparts);
}

/* ---------------------------------------------------------------- q3: filter + 2 hash joins + group-by + sort
 * reference benchmarks/queries/q3.sql (physical shape: benchmarks/src/bin/tpch.rs:286-351 runs it through DataFusion's
 * HashJoinExec(CollectLeft) x 2 -> AggregateExec -> SortExec) over Arrow-physical columns:
 *   customer(c_custkey i64, c_mktsegment Utf8)  orders(o_orderkey, o_custkey i64, o_orderdate date32, o_shippriority i32)
 *   lineitem(l_orderkey i64, l_extendedprice / l_discount Decimal128(15,2), l_shipdate date32)
 * J1 build: customers of `segment`; J1 probe: orders with o_orderdate < date; J2 build: the J1 output keyed by o_orderkey;
 * J2 probe: lineitem with l_shipdate > date; aggregate: SUM(l_extendedprice * (1 - l_discount)) as Decimal128(38,4) by
 * (l_orderkey, o_orderdate, o_shippriority); order by revenue DESC, o_orderdate ASC (ties: l_orderkey ASC, so that the
 * output is a pure function of the input).  Executed the way a vectorised multi-core CPU engine does it: chained hash
 * tables with key re-verification [UPSTREAM-KNOWLEDGE], the probe side split over threads, per-thread partial aggregates
 * merged at the end.  Returns the number of groups; the first `cap` of them are written. */
typedef struct { i64 okey; int32_t odate, oprio; i128 rev; int used; } q3_group;
typedef struct { q3_group* g; u64 mask; i64 n; } q3_table;
static void q3_table_init(q3_table* t, u64 slots) { t->g = (q3_group*)calloc(slots, sizeof(q3_group)); t->mask = slots - 1; t->n = 0; }
static void q3_table_add(q3_table* t, i64 okey, int32_t odate, int32_t oprio, i128 rev);
static void q3_table_grow(q3_table* t) {
  q3_table b; q3_table_init(&b, (t->mask + 1) * 2);
  for (u64 s = 0; s <= t->mask; ++s) if (t->g[s].used) q3_table_add(&b, t->g[s].okey, t->g[s].odate, t->g[s].oprio, t->g[s].rev);
  free(t->g); *t = b;
}
static void q3_table_add(q3_table* t, i64 okey, int32_t odate, int32_t oprio, i128 rev) {
  if ((u64)t->n * 2 > t->mask) q3_table_grow(t);
  u64 s = mix64((u64)okey ^ ((u64)(uint32_t)odate << 32) ^ (u64)(uint32_t)oprio * 0x9E3779B97F4A7C15ull) & t->mask;
  for (;; s = (s + 1) & t->mask) {
    q3_group* q = &t->g[s];
    if (!q->used) { q->used = 1; q->okey = okey; q->odate = odate; q->oprio = oprio; q->rev = rev; t->n++; return; }
    if (q->okey == okey && q->odate == odate && q->oprio == oprio) { q->rev += rev; return; }
  }
}
static int q3_cmp(const void* a, const void* b) {
  const q3_group* x = (const q3_group*)a; const q3_group* y = (const q3_group*)b;
  if (x->rev != y->rev) return x->rev > y->rev ? -1 : 1;
  if (x->odate != y->odate) return x->odate < y->odate ? -1 : 1;
  return x->okey < y->okey ? -1 : (x->okey > y->okey ? 1 : 0);
}

i64 oracle_q3(i64 n_cust, const i64* c_custkey, const uint8_t* c_mktsegment, const int32_t* c_seg_off, const char* segment,
              i64 n_orders, const i64* o_orderkey, const i64* o_custkey, const int32_t* o_orderdate, const int32_t* o_shippriority,
              i64 n_line, const i64* l_orderkey, const u64* l_extendedprice, const u64* l_discount, const int32_t* l_shipdate, int32_t date,
              i64 cap, i64* out_orderkey, u64* out_revenue /*[cap][2]*/, int32_t* out_orderdate, int32_t* out_shippriority,
              i64* stats /* optional [4]: J1 build rows, J1 output rows, J2 probe rows (pass the filter), J2 matches */) {
  const int nthreads = oracle_num_threads();
  const size_t seglen = strlen(segment);
  /* J1 build side: FilterExec(c_mktsegment = segment) -> keys */
  i64* ck = (i64*)malloc((size_t)(n_cust > 0 ? n_cust : 1) * 8); i64 nck = 0;
  for (i64 i = 0; i < n_cust; ++i) {
    const int32_t a = c_seg_off[i], b = c_seg_off[i + 1];
    if ((size_t)(b - a) == seglen && memcmp(c_mktsegment + a, segment, seglen) == 0) ck[nck++] = c_custkey[i];
  }
  void* t1 = oracle_join_build(ck, nck);
  const oracle_join_table* T1 = (const oracle_join_table*)t1;
  /* J1 probe: FilterExec(o_orderdate < date) -> inner join on custkey; output rows keep the probe (orders) order */
  i64* cnt = (i64*)calloc((size_t)nthreads + 1, 8);
  i64* jk = NULL; int32_t* jd = NULL; int32_t* jp = NULL; i64 nj = 0;
  for (int pass = 0; pass < 2; ++pass) {
#pragma omp parallel
    {
      int tid = 0;
#ifdef _OPENMP
      tid = omp_get_thread_num();
#endif
      const i64 a = n_orders * tid / nthreads, b = n_orders * (tid + 1) / nthreads;
      i64 w = pass ? cnt[tid] : 0;
      for (i64 j = a; j < b; ++j) {
        if (!(o_orderdate[j] < date)) continue;
        const i64 k = o_custkey[j];
        for (uint32_t r = T1->head[mix64((u64)k) & T1->mask]; r != 0xFFFFFFFFu; r = T1->next[r])
          if (T1->keys[r] == k) { if (pass) { jk[w] = o_orderkey[j]; jd[w] = o_orderdate[j]; jp[w] = o_shippriority[j]; } ++w; }
      }
      if (!pass) cnt[tid + 1] = w;
    }
    if (!pass) {
      cnt[0] = 0; for (int t = 0; t < nthreads; ++t) cnt[t + 1] += cnt[t];
      nj = cnt[nthreads];
      jk = (i64*)malloc((size_t)(nj > 0 ? nj : 1) * 8); jd = (int32_t*)malloc((size_t)(nj > 0 ? nj : 1) * 4); jp = (int32_t*)malloc((size_t)(nj > 0 ? nj : 1) * 4);
    }
  }
  oracle_join_free(t1);
  /* J2: build on o_orderkey of the J1 output, probe FilterExec(l_shipdate > date), aggregate per thread */
  void* t2 = oracle_join_build(jk, nj);
  const oracle_join_table* T2 = (const oracle_join_table*)t2;
  q3_table* parts = (q3_table*)calloc((size_t)nthreads, sizeof(q3_table));
  i64 probes = 0, matches = 0;
#pragma omp parallel reduction(+ : probes, matches)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    q3_table* pt = &parts[tid]; q3_table_init(pt, 1024);
    const i64 a = n_line * tid / nthreads, b = n_line * (tid + 1) / nthreads;
    for (i64 i = a; i < b; ++i) {
      if (!(l_shipdate[i] > date)) continue;
      ++probes;
      const i64 k = l_orderkey[i];
      for (uint32_t r = T2->head[mix64((u64)k) & T2->mask]; r != 0xFFFFFFFFu; r = T2->next[r]) {
        if (T2->keys[r] != k) continue;
        const i128 ext = (i128)(((u128)l_extendedprice[2 * i + 1] << 64) | l_extendedprice[2 * i]);
        const i128 disc = (i128)(((u128)l_discount[2 * i + 1] << 64) | l_discount[2 * i]);
        q3_table_add(pt, k, jd[r], jp[r], ext * (100 - disc));      /* Decimal(15,2) * (Decimal(20,0) - Decimal(15,2)) -> scale 4 */
        ++matches;
      }
    }
  }
  oracle_join_free(t2);
  q3_table all; q3_table_init(&all, 1024);
  for (int t = 0; t < nthreads; ++t) {
    for (u64 s = 0; s <= parts[t].mask; ++s) if (parts[t].g[s].used) q3_table_add(&all, parts[t].g[s].okey, parts[t].g[s].odate, parts[t].g[s].oprio, parts[t].g[s].rev);
    free(parts[t].g);
  }
  free(parts);
  q3_group* out = (q3_group*)malloc((size_t)(all.n > 0 ? all.n : 1) * sizeof(q3_group)); i64 ng = 0;
  for (u64 s = 0; s <= all.mask; ++s) if (all.g[s].used) out[ng++] = all.g[s];
  qsort(out, (size_t)ng, sizeof(q3_group), q3_cmp);
  for (i64 g = 0; g < ng && g < cap; ++g) {
    out_orderkey[g] = out[g].okey; out_revenue[2 * g] = (u64)out[g].rev; out_revenue[2 * g + 1] = (u64)((u128)out[g].rev >> 64);
    out_orderdate[g] = out[g].odate; out_shippriority[g] = out[g].oprio;
  }
  if (stats) { stats[0] = nck; stats[1] = nj; stats[2] = probes; stats[3] = matches; }
  free(out); free(all.g); free(ck); free(cnt); free(jk); free(jd); free(jp);
  return ng;
}

/* ---------------------------------------------------------------- q5: six-way join + group-by + sort
 * reference benchmarks/queries/q5.sql: region(r_name = 'ASIA') |x| nation |x| customer |x| orders(1994) |x| lineitem |x| supplier
 * on (l_suppkey = s_suppkey AND c_nationkey = s_nationkey), SUM(l_extendedprice * (1 - l_discount)) by n_name, ORDER BY revenue
 * DESC (tpch.rs:286-351 runs it as five HashJoinExec(CollectLeft) -> AggregateExec -> SortExec).  Executed here the way
 * oracle_q3 is: chained hash tables with key re-verification, the probe sides split over threads, per-thread partial sums.
 * nation_region[k] = region key of nation key k (n_nations entries); groups are nation KEYS (the caller maps them to n_name).
 * Returns the number of groups; out_nation / out_revenue ([..][2], Decimal128(38,4) as lo,hi) ordered by revenue DESC, nation key. */
typedef struct { int32_t nation; i128 rev; } q5_group;
static int q5_cmp(const void* a, const void* b) {
  const q5_group* x = (const q5_group*)a; const q5_group* y = (const q5_group*)b;
  if (x->rev != y->rev) return x->rev > y->rev ? -1 : 1;
  return x->nation < y->nation ? -1 : (x->nation > y->nation ? 1 : 0);
}
i64 oracle_q5(i64 n_nations, const i64* nation_region, i64 region_key,
              i64 n_cust, const i64* c_custkey, const i64* c_nationkey,
              i64 n_orders, const i64* o_orderkey, const i64* o_custkey, const int32_t* o_orderdate, int32_t date_lo, int32_t date_hi,
              i64 n_line, const i64* l_orderkey, const i64* l_suppkey, const u64* l_extendedprice, const u64* l_discount,
              i64 n_supp, const i64* s_suppkey, const i64* s_nationkey,
              int32_t* out_nation, u64* out_revenue, i64* stats /* optional [3]: customers kept, orders kept, lineitem pairs that reach the aggregate */) {
  const int nthreads = oracle_num_threads();
  if (n_nations > 64) return -1;
  /* region |x| nation |x| customer: customers of the region's nations */
  i64* ck = (i64*)malloc((size_t)(n_cust > 0 ? n_cust : 1) * 8); int32_t* cn = (int32_t*)malloc((size_t)(n_cust > 0 ? n_cust : 1) * 4); i64 nck = 0;
  for (i64 i = 0; i < n_cust; ++i) {
    const i64 nk = c_nationkey[i];
    if (nk >= 0 && nk < n_nations && nation_region[nk] == region_key) { ck[nck] = c_custkey[i]; cn[nck] = (int32_t)nk; ++nck; }
  }
  void* t1 = oracle_join_build(ck, nck);
  const oracle_join_table* T1 = (const oracle_join_table*)t1;
  /* |x| orders of the date range: (o_orderkey, customer's nation), in orders order */
  i64* cnt = (i64*)calloc((size_t)nthreads + 1, 8);
  i64* jk = NULL; int32_t* jn = NULL; i64 nj = 0;
  for (int pass = 0; pass < 2; ++pass) {
#pragma omp parallel
    {
      int tid = 0;
#ifdef _OPENMP
      tid = omp_get_thread_num();
#endif
      const i64 a = n_orders * tid / nthreads, b = n_orders * (tid + 1) / nthreads;
      i64 w = pass ? cnt[tid] : 0;
      for (i64 j = a; j < b; ++j) {
        if (!(o_orderdate[j] >= date_lo && o_orderdate[j] < date_hi)) continue;
        const i64 k = o_custkey[j];
        for (uint32_t r = T1->head[mix64((u64)k) & T1->mask]; r != 0xFFFFFFFFu; r = T1->next[r])
          if (T1->keys[r] == k) { if (pass) { jk[w] = o_orderkey[j]; jn[w] = cn[r]; } ++w; }
      }
      if (!pass) cnt[tid + 1] = w;
    }
    if (!pass) {
      cnt[0] = 0; for (int t = 0; t < nthreads; ++t) cnt[t + 1] += cnt[t];
      nj = cnt[nthreads];
      jk = (i64*)malloc((size_t)(nj > 0 ? nj : 1) * 8); jn = (int32_t*)malloc((size_t)(nj > 0 ? nj : 1) * 4);
    }
  }
  oracle_join_free(t1);
  /* |x| lineitem on the order key, |x| supplier on (suppkey, nation); partial sums per thread */
  void* t2 = oracle_join_build(jk, nj);
  void* t3 = oracle_join_build(s_suppkey, n_supp);
  const oracle_join_table* T2 = (const oracle_join_table*)t2; const oracle_join_table* T3 = (const oracle_join_table*)t3;
  i128* part = (i128*)calloc((size_t)nthreads * 64, sizeof(i128));
  i64 pairs = 0;
#pragma omp parallel reduction(+ : pairs)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    i128* mine = part + (size_t)tid * 64;
    const i64 a = n_line * tid / nthreads, b = n_line * (tid + 1) / nthreads;
    for (i64 i = a; i < b; ++i) {
      const i64 k = l_orderkey[i];
      for (uint32_t r = T2->head[mix64((u64)k) & T2->mask]; r != 0xFFFFFFFFu; r = T2->next[r]) {
        if (T2->keys[r] != k) continue;
        const i64 sk = l_suppkey[i];
        for (uint32_t q = T3->head[mix64((u64)sk) & T3->mask]; q != 0xFFFFFFFFu; q = T3->next[q]) {
          if (T3->keys[q] != sk || s_nationkey[q] != (i64)jn[r]) continue;
          const i128 ext = (i128)(((u128)l_extendedprice[2 * i + 1] << 64) | l_extendedprice[2 * i]);
          const i128 disc = (i128)(((u128)l_discount[2 * i + 1] << 64) | l_discount[2 * i]);
          mine[jn[r]] += ext * (100 - disc);
          ++pairs;
        }
      }
    }
  }
  oracle_join_free(t2); oracle_join_free(t3);
  /* a nation is a group when at least one row reached it: keep a presence flag next to the sums */
  q5_group out[64]; i64 ng = 0;
  for (int k = 0; k < (int)n_nations; ++k) {
    i128 s = 0; for (int t = 0; t < nthreads; ++t) s += part[(size_t)t * 64 + k];
    if (s != 0) { out[ng].nation = k; out[ng].rev = s; ++ng; }      /* (revenue of a non-empty group is positive: prices > 0, discounts < 1) */
  }
  qsort(out, (size_t)ng, sizeof(q5_group), q5_cmp);
  for (i64 g = 0; g < ng; ++g) { out_nation[g] = out[g].nation; out_revenue[2 * g] = (u64)out[g].rev; out_revenue[2 * g + 1] = (u64)((u128)out[g].rev >> 64); }
  if (stats) { stats[0] = nck; stats[1] = nj; stats[2] = pairs; }
  free(part); free(ck); free(cn); free(cnt); free(jk); free(jn);
  return ng;
}
