// Internal C++ launch interface between the C ABI (capi.cpp) and the HIP kernels.
#pragma once
#include "gpuq_dev.h"

namespace gpuq {

// ----- project / materialise
constexpr int MAX_OUTS = 16;
struct OutCol {
  void* data;          // fixed-width destination (4/8/16 B per row by cls; CC_STR stores the 16-B packed form)
  u64* validity;       // optional validity bitmap as 64-bit words (bit i of word w = row 64*w+i); nullptr = not written
  int32_t reg;
  int32_t cls;         // CC_I32 / CC_U32 / CC_I64 / CC_I128 / CC_STR / CC_BIT(data = u64 bitmap words)
};
struct OutSpec { int32_t n_out; int32_t pad; OutCol cols[MAX_OUTS]; };

// ----- aggregate
constexpr int MAX_ACCS = 12;
constexpr int MAX_KEYS = 4;
enum AccKind : int32_t { ACC_SUM = 0, ACC_COUNT = 1, ACC_COUNT_STAR = 2, ACC_MIN = 3, ACC_MAX = 4, ACC_FSUM = 5, ACC_FMIN = 6, ACC_FMAX = 7 };
struct AggSpec {
  int32_t n_keys, n_accs;
  int32_t key_reg[MAX_KEYS];
  int32_t acc_kind[MAX_ACCS];
  int32_t acc_reg[MAX_ACCS];
};

// Result layout shared by the tiny and hash aggregates (device memory, caller allocated):
//   keys  [cap][n_keys] as (lo,hi) u64 pairs      key_nulls[cap] bitmask over keys
//   cells [cap][n_accs] as (lo,hi) u64 pairs      n_groups (device u32)
struct AggOut {
  u64* keys; uint32_t* key_nulls; u64* cells; uint32_t* n_groups; int32_t cap; int32_t pad;
};

// scan kernels (kernels_scan.hip)


// hash kernels (kernels_hash.hip)
constexpr int MAX_KW = 2 * MAX_KEYS + 1;
struct KeySpec {
  int32_t n_keys;
  int32_t key_reg[MAX_KEYS];
  int32_t key_wide[MAX_KEYS];  // 1: key needs both 64-bit halves (Decimal128, packed Utf8); 0: hi is the sign extension
  int32_t null_word;           // 1: append the key null-mask as a key word (group-by keys, null_equals_null joins)
  int32_t key_words;           // total u64 key words per slot
  int32_t word_reg[MAX_KW];    // key word q comes from register word_reg[q] ...
  int32_t word_half[MAX_KW];   // ... half 0 = lo, 1 = hi, 2 = the key null mask
};
// Open-addressing table, linear probing, one slot = slot_words u64:
//   [0]            low 32 bits state (0 empty, 1 locked, else hash tag|2); high 32 bits payload (join: chain head row)
//   [1..key_words] key words
//   [1+key_words..] aggregate cells, two u64 (lo,hi) per accumulator
// Join tables over ONE narrow (<= 64-bit) integer key whose values span a bounded range use direct addressing instead:
// dense[key - dense_min] = chain head row (0xFFFFFFFF = no such key), dense_range entries.  One load per probe, no probing
// loop, and a probe side that is clustered / sorted by the key (fact-table foreign keys) walks the table sequentially.
struct HashTable {
  u64* slots;
  u64 n_slots;         // power of two
  int32_t slot_words;
  int32_t key_words;
  uint32_t* dense;     // nullptr = open addressing (slots)
  i64 dense_min;
  u64 dense_range;
  // Sparse domains (few keys in a wide range: a filtered dimension table) add a presence bitmap, one bit per key value, and leave
  // the row array uninitialised: only bits are cleared at build time (range / 8 bytes instead of 4 x range), a probe that misses
  // touches the bitmap alone (32 x denser than the array: L2- or Infinity-Cache-resident), a hit reads bitmap + array.
  // Only for unique build keys (duplicates need the chain heads initialised).  nullptr = the array alone (NIL = no key).
  uint32_t* dense_bits;
};
// chain fusion: the build kernel first looks its rows up in ANOTHER join's table (kernels_hash.hip k_join_build_body)
struct SemiProbe { HashTable T; KeySpec K; int32_t null_eq; int32_t on; uint32_t* hit_out; u64* rows_out; };
enum JoinType : int32_t { JT_INNER = 0, JT_LEFT = 1, JT_RIGHT = 2, JT_FULL = 3, JT_LEFT_SEMI = 4, JT_LEFT_ANTI = 5, JT_RIGHT_SEMI = 6, JT_RIGHT_ANTI = 7 };

// build: payload = payload_via ? via[payload_via-1][pos] : pos.  next == nullptr => unique keys only (FLAG_DUP_BUILD_KEY on a duplicate)

// sort / partition (kernels_sort.hip)
constexpr int MAX_SORT_KEYS = 4;
struct SortSpec {
  int32_t n_keys;
  int32_t reg[MAX_SORT_KEYS];
  int32_t desc[MAX_SORT_KEYS];
  int32_t nulls_first[MAX_SORT_KEYS];
  int32_t kind[MAX_SORT_KEYS];     // 0 integer/decimal/date, 1 float64 (total order), 2 packed Utf8
};
constexpr int SORT_MAX_PASSES = 16;  // 8-bit digits of a composite key of up to 128 bits
struct SortPack {                   // computed on the host from the per-key min/max
  u64 base_lo[MAX_SORT_KEYS], base_hi[MAX_SORT_KEYS];   // min (ASC) or max (DESC) in the ordered view
  int32_t shift[MAX_SORT_KEYS];     // bit position of the key's field in the composite
  int32_t null_bit[MAX_SORT_KEYS];  // bit (inside the field) of the null flag, -1 = none
  int32_t rshift[MAX_SORT_KEYS];    // packed Utf8: drop the length byte and the bytes beyond the longest string
  int32_t vbits[MAX_SORT_KEYS];     // width of the value field (without the null flag)
  int32_t check;                    // 1: the layout was GUESSED from a sample -- the pack kernel verifies every row against it and raises
                                    // hist[SORT_MAX_PASSES * 256] when a value falls outside its field or a NULL meets a key without a null bit
                                    // 2: the same, reported as FLAG_SORT_LAYOUT in the operator's status word (deferred execution: nobody reads hist)
  int32_t total_bits;               // width of the composite key: padding records (rows beyond a device-side row count) hold all ones in it
};

// aggregate post-processing (kernels_scan.hip): AoS result -> one (lo,hi) column per key / accumulator
struct AggSoA {
  ulonglong2* key_col[MAX_KEYS]; u64* key_valid[MAX_KEYS];   // validity words, bit g of word g/64
  ulonglong2* acc_col[MAX_ACCS];
};

// generator (kernels_gen.hip)


// deferred execution: up to 64 device words gathered into one contiguous block (gpuq_ops_settle)
struct GatherWords { int32_t n; int32_t pad; const u64* src[64]; };

#ifndef GPUQ_JIT
// ---- host launch interface (AOT build only)
void launch_gather_words(hipStream_t s, const GatherWords& g, u64* out);
void set_num_cus(int n);
int num_cus();
void launch_filter_bitmap(hipStream_t s, const DevProgram& P, i64 n, u64* bitmap, uint32_t* block_counts, int nblocks, i64 words_per_block);
void launch_scan_block_counts(hipStream_t s, uint32_t* block_counts, int nblocks, u64* total_out);
void launch_compact(hipStream_t s, const u64* bitmap, const uint32_t* block_offsets, int nblocks, i64 words_per_block, i64 n,
                    const uint32_t* sel_in, uint32_t* sel_out);
void launch_project(hipStream_t s, const DevProgram& P, i64 n, const OutSpec& O);
int agg_tiny_max_groups(int n_accs);
size_t agg_tiny_workspace_bytes(int gmax, int n_keys, int n_accs, int* nblocks_out);
void launch_agg_tiny(hipStream_t s, const DevProgram& P, i64 n, const AggSpec& A, int gmax, void* workspace);
void launch_agg_tiny_merge(hipStream_t s, const DevProgram& P, i64 n, const AggSpec& A, int gmax, void* workspace, const AggOut& out);
void launch_ht_init(hipStream_t s, const HashTable& T, const AggSpec* A);
void launch_agg_hash(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const AggSpec& A, const HashTable& T);
void launch_key_sample(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, i64 stride, i64 nsample, uint32_t* bitmap, u64 nbits, unsigned long long* passed);
uint32_t agg_lds_slots(const HashTable& T);
int agg_lds_grid(i64 n);
void launch_agg_lds(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const AggSpec& A, const HashTable& T, u64* fstage, int n_fsum);
void launch_agg_bucket_id(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, u64 bucket_mask, u64* bid, uint32_t* ids);
void launch_bucket_bounds(hipStream_t s, const u64* sorted_bid, i64 n, u64 nbuckets, uint32_t* bounds, int shift);
void launch_agg_bucket(hipStream_t s, const DevProgram& P, const KeySpec& K, const AggSpec& A, const uint32_t* ids, const uint32_t* bounds, uint32_t nbuckets,
                       uint32_t cap, int slot_words, const AggOut& out);
void launch_agg_hash_extract(hipStream_t s, const KeySpec& K, const AggSpec& A, const HashTable& T, const AggOut& out, uint32_t* flags);
// false: no interpreter kernel for this shape (chain fusion over more than 8 input columns)
bool launch_join_build(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, uint32_t* next, uint32_t* present,
                       int payload_via, int null_equals_null, const SemiProbe* semi = nullptr);
// key range of the rows a join build would insert: out = {min (i64), max (i64), count (u64)}, pre-set by the caller to {INT64_MAX, INT64_MIN, 0}
void launch_join_keyrange(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, int null_equals_null, u64* out, i64 wstep = 1);   // wstep > 1: every wstep-th 64-row word
// unique build keys: every wave owns `wpw` consecutive 64-row words (segment g = words [g*wpw, (g+1)*wpw)) and writes its pairs, in probe
// order, to seg_build / seg_probe starting at row g*wpw*64; seg_counts[g] = pairs of the segment.  launch_copy_segments then moves
// the segments to their final places (seg_counts already scanned to exclusive offsets).
void launch_join_probe_unique(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, int join_type, int null_equals_null, int payload_via,
                              uint32_t* seg_build, uint32_t* seg_probe, uint32_t* seg_counts, int nsegs, i64 wpw, uint32_t* visited);
void launch_copy_segments(hipStream_t s, const uint32_t* seg_build, const uint32_t* seg_probe, const uint32_t* seg_offsets, int nsegs, i64 wpw, i64 n,
                          const u64* total, uint32_t* out_build, uint32_t* out_probe, u64 out_cap, uint32_t* flags);
// partitioned probe over a direct-addressed table (kernels_hash.hip): records = (table index << 32 | probe row), one scatter pass on
// the high index bits; offsets of the dropped-rows digit (hist[nparts * nblocks]) = number of live records after the scan
void launch_join_locality(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, i64 stride, i64 nsample, u64* out);
struct RjGeomHost { uint32_t nparts, shift; i64 tile; int32_t nblocks; };
void rj_geometry(i64 n, u64 range, int slice_log2, RjGeomHost* g);
size_t rj_hist_entries(const RjGeomHost& g);
void launch_rj_partition(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, int payload_via, const RjGeomHost& g,
                         u64* rec, u64* rec_out, int32_t* hist, void* scan_ws, size_t scan_ws_bytes);
int rj_probe_geometry(i64 n, i64* wpw_out);
void launch_rj_probe(hipStream_t s, const u64* rec, const int32_t* n_live, const HashTable& T, int join_type, uint32_t* seg_build, uint32_t* seg_probe,
                     uint32_t* seg_counts, int nblocks, i64 wpw);
void launch_bitmap_select(hipStream_t s, const u64* present, const u64* visited, int matched, i64 nwords, i64 n, u64* bitmap,
                          uint32_t* block_counts, int nblocks, i64 wpb);
void launch_join_probe(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, const HashTable& T, const uint32_t* next,
                       int join_type, int payload_via, int null_equals_null, uint32_t* out_build, uint32_t* out_probe,
                       u64 out_cap, u64* out_count, uint32_t* visited);
int sort_minmax_blocks(i64 n);
void launch_sort_minmax(hipStream_t s, const DevProgram& P, i64 n, const SortSpec& S, u64* out, int nblocks, i64 wstep = 1);   // wstep > 1: every wstep-th 64-row word only (a sample)
int sort_max_passes();
void launch_sort_pack(hipStream_t s, const DevProgram& P, i64 n, const SortSpec& S, const SortPack& K, u64* key_lo, u64* key_hi, uint32_t* ids, u64* hist, int hist_passes);
void launch_part_pid(hipStream_t s, const DevProgram& P, i64 n, const KeySpec& K, uint32_t nparts, u64* pid_out, uint32_t* ids);
void launch_part_offsets(hipStream_t s, const u64* pid, i64 n, uint32_t nparts, uint32_t* counts_ws, u64* offsets_out, int shift);
void launch_counts_to_ghist(hipStream_t s, const uint32_t* counts, uint32_t np, u64* ghist);
void launch_gather_u64(hipStream_t s, const u64* src, const uint32_t* idx, i64 n, u64* dst);
int sort_small_max();
int sort_direct_max();
void launch_sort_direct(hipStream_t s, const DevProgram& P, i64 n, const SortSpec& S, uint32_t* perm);
void launch_sort_small(hipStream_t s, const u64* klo, const u64* khi, const uint32_t* ids, i64 n, uint32_t* out);
size_t onesweep_ws_bytes(i64 n);
int onesweep_max_passes();
size_t merge_splits_entries(i64 max_len, int n_pairs);
void launch_merge_pairs(hipStream_t s, const u64* klo, const u64* khi, const uint32_t* ids, const i64* pairs, int n_pairs, i64 max_len, i64* splits,
                        u64* klo_out, u64* khi_out, uint32_t* ids_out);
void launch_radix_ghist(hipStream_t s, const u64* keys, i64 n, int shift0, int npasses, u64* ghist);
void launch_onesweep_pass(hipStream_t s, const u64* keys, const uint32_t* vals, i64 n, int shift, u64* gexcl, void* ws, size_t ws_bytes,
                          u64* keys_out, uint32_t* vals_out, int ids_only /* 0 records, 1 row ids only, 2 packed records + row ids */);
void launch_agg_emit(hipStream_t s, const AggOut& raw, int n_keys, int n_accs, uint32_t n_groups, const AggSoA& soa, const uint32_t* n_groups_dev = nullptr);
void launch_concat_bitmap(hipStream_t s, u64* dst, i64 dst_bit_offset, const uint8_t* src, i64 src_bit_offset, i64 n_bits);
void launch_unpack_utf8_lengths(hipStream_t s, const ulonglong2* packed, i64 n, int32_t* lens_out, uint32_t* too_long);
constexpr int LIKE_MAX_TOKENS = 256;
constexpr int LIKE_MAX_SEGS = 8;
struct LikePattern {
  int32_t n; int32_t regex_mode; uint16_t tok[LIKE_MAX_TOKENS];     // 0..255 literal byte, 256 '_', 257 '%'
  // the pattern as literal segments between '%' (n_seg < 0: it holds '_' or too many segments: the general matcher runs)
  int32_t n_seg, anchored_start, anchored_end, pad;
  int32_t seg_off[LIKE_MAX_SEGS], seg_len[LIKE_MAX_SEGS];
  unsigned long long seg_first8[LIKE_MAX_SEGS], seg_mask8[LIKE_MAX_SEGS];
};
void launch_like_utf8(hipStream_t s, const uint8_t* data, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n, const LikePattern& pat, int negated,
                      u64* bits_out, u64* valid_out);
void launch_mark_rows(hipStream_t s, const uint32_t* rows, i64 n, uint8_t* bitmap);
void launch_utf8_compare(hipStream_t s, const uint8_t* adata, const int32_t* aoffs, const uint8_t* avalid, const uint32_t* aidx, const uint8_t* bdata, const int32_t* boffs,
                         const uint8_t* bvalid, const uint32_t* bidx, int32_t blen, i64 n, int op, u64* bits_out, u64* valid_out);
void launch_sort_decode(hipStream_t s, const u64* recs, int rec_shift, i64 n, const SortPack& K, int desc, int nulls_first, int width, void* out, u64* valid_out);
void launch_cross_pairs(hipStream_t s, i64 n_left, i64 n_right, uint32_t* left_rows, uint32_t* right_rows);
void launch_offsets_rebase(hipStream_t s, const int32_t* src, i64 n, int32_t delta, int32_t* dst);
void launch_take_utf8_lengths(hipStream_t s, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n, int32_t* lens, u64* valid_out);
void launch_take_utf8_bytes(hipStream_t s, const uint8_t* data, const int32_t* offsets, const uint32_t* idx, i64 n, const int32_t* out_offsets, uint8_t* out, i64 total_bytes);
void launch_unpack_utf8_bytes(hipStream_t s, const ulonglong2* packed, i64 n, const int32_t* offsets, uint8_t* data_out);
void launch_exclusive_scan_i32(hipStream_t s, int32_t* data, i64 n, void* workspace, size_t ws_bytes);   // in place, n+1 entries out
size_t exclusive_scan_ws_bytes(i64 n);
// ----- LZ4 block codec (shuffle sink / source, kernels_lz4.hip)
struct Lz4Block { int64_t src; int64_t slot; int32_t len; int32_t pad; };     // compress: one <= 64 KiB input block and its worst-case output slot
constexpr int LZ4_UNIT_BLOCK = 0, LZ4_UNIT_STORED = 1, LZ4_UNIT_FRAME_BLOCKS = 2;
constexpr int LZ4_UNIT_BLOCK_CHECKSUM = 1;
struct Lz4Unit { int64_t src, dst, src_len, dst_len; int32_t mode, flags; };  // decompress: one independent block, or the blocks of one linked frame
constexpr int64_t LZ4_BLOCK_BYTES = 65536;
inline int64_t lz4_slot_bytes(int64_t len) { return ((len + len / 255 + 16) + 15) & ~(int64_t)15; }
void launch_lz4_compress(hipStream_t s, const uint8_t* src, uint8_t* slots, const Lz4Block* blocks, int n_blocks, int32_t* csize);
void launch_lz4_layout(hipStream_t s, const Lz4Block* blocks, const int32_t* csize, const int32_t* buf_first_block, int n_buffers, int64_t* buf_off, int64_t* buf_len,
                       int64_t* blk_dst);
void launch_lz4_pack(hipStream_t s, const uint8_t* src, const uint8_t* slots, const Lz4Block* blocks, int n_blocks, const int32_t* csize, const int32_t* blk_buffer,
                     const int32_t* buf_first_block, const int64_t* buf_off, const int64_t* buf_len, const int64_t* blk_dst, uint8_t* body);
void launch_lz4_decode(hipStream_t s, const uint8_t* src, int64_t src_bytes, uint8_t* dst, const Lz4Unit* units, int n_units, uint32_t* status);
struct Utf8Piece { const int32_t* tmp; int64_t n; int32_t* dst; int64_t at, dl, start, used; int32_t first, pad; };   // one batch of a Utf8 column (kernels_lz4.hip)
void launch_utf8_piece_starts(hipStream_t s, Utf8Piece* pieces, int n_pieces, uint32_t* gap);
void launch_utf8_piece_offsets(hipStream_t s, const Utf8Piece* pieces, int n_pieces, int64_t max_rows);
void launch_utf8_piece_validate(hipStream_t s, const Utf8Piece* pieces, int n_pieces, int64_t max_rows, uint32_t* status);
void launch_utf8_piece_compact(hipStream_t s, const Utf8Piece* pieces, int n_pieces, int64_t max_bytes, const uint8_t* from, uint8_t* to);
// one Parquet page, or one block of a linked LZ4 frame (kernels_lz4.hip).  mode: 0 stored, 1 Snappy, 2 LZ4 block of a linked frame, 3 stored block of a
// linked frame (its bytes can be the source of later blocks' matches).  s_off / c_off: the job's place in the resolve array / the per-position arrays;
// linked frames: f_off = the FRAME's place in the resolve array and p_base = the block's position in the frame's output (copies may reach back
// across blocks; s_off = f_off + p_base), elsewhere f_off = s_off and p_base = 0
struct UnpackJob { int64_t src, dst, src_len, dst_len, raw_prefix; int32_t mode, pad; int64_t s_off, c_off, f_off, p_base; };
void launch_unpack_pages(hipStream_t s, const uint8_t* src, uint8_t* dst, const UnpackJob* jobs, int n_jobs, uint32_t* status);
// mode 4 = ZSTD frames (kernels_zstd_dev.hip + zstd_launch.cpp; the unpack kernels copy such a job's raw prefix and leave the rest to this launch);
// `which`: indices of the mode-4 jobs, `scratch`: zstd_scratch_bytes(n) bytes
size_t zstd_scratch_bytes(int n_pages);
void launch_zstd_pages(hipStream_t s, const uint8_t* src, uint8_t* dst, const UnpackJob* jobs, const int32_t* which, int n, uint8_t* scratch, uint32_t* status);
// Snappy without a serial element walk (kernels_lz4.hip).  resolve: one 32-bit word per uncompressed byte of the Snappy jobs (s_off); blkmap: (job, first
// byte) per 4096 of them.  jump_a / jump_b / olen (32-bit) and mark (8-bit): one entry per compressed byte + one per job (c_off; c_slots in all, + 1 for the
// scan); cmap: (job, first position) per 4096 of those.  counts: (mark_rounds + rounds + 3) * n_jobs zeroed words.
struct SnappyPjBuffers {
  uint32_t* resolve; const uint2* blkmap; int n_blocks; int rounds;
  uint32_t* jump_a; uint32_t* jump_b; uint32_t* olen; uint8_t* mark; const uint2* cmap; int n_cblocks; int mark_rounds; int64_t c_slots;
  void* scan_ws; size_t scan_ws_bytes; uint32_t* counts;
};
void launch_unpack_pages_pj(hipStream_t s, const uint8_t* src, uint8_t* dst, const UnpackJob* jobs, int n_jobs, const SnappyPjBuffers& B, uint32_t* status);
void launch_utf8_code_rows(hipStream_t s, const void* codes, int width, const uint8_t* validity, i64 n, uint32_t* rows);      // width 8 (Int64) or 4 (UInt32)
void launch_utf8_sort_piece(hipStream_t s, const uint8_t* data, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n, int piece, void* out, u64* valid_out);
void launch_utf8_max_len(hipStream_t s, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n, int32_t* out);
void launch_utf8_intern(hipStream_t s, const uint8_t* ddata, const int32_t* doffs, const uint8_t* data, const int32_t* offsets, const uint8_t* validity, const uint32_t* idx, i64 n,
                        u64* table, u64 mask, int insert, i64* codes, u64* valid_out, uint32_t* flags);
void launch_popcount_bits(hipStream_t s, const uint8_t* bits, int64_t n_bits, unsigned long long* out);

// ----- scan-side decode (kernels_scanfmt.hip): delimited text and Parquet pages
enum CsvKind : int32_t { CSV_SKIP = 0, CSV_I32 = 1, CSV_I64 = 2, CSV_DATE32 = 3, CSV_DEC128 = 4, CSV_F64 = 5, CSV_BOOL = 6, CSV_UTF8 = 7 };
constexpr int CSV_MAX_FIELDS = 64;
struct CsvSpec {
  int32_t n_fields; int32_t kind[CSV_MAX_FIELDS]; int32_t out[CSV_MAX_FIELDS]; int32_t scale[CSV_MAX_FIELDS]; int32_t nullable[CSV_MAX_FIELDS];
  int32_t prec[CSV_MAX_FIELDS];
  uint8_t delim, quote; uint8_t pad[2];
};
struct CsvOut { void* data[CSV_MAX_FIELDS]; u64* valid[CSV_MAX_FIELDS]; uint32_t* str_start[CSV_MAX_FIELDS]; int32_t* str_len[CSV_MAX_FIELDS]; };
constexpr uint32_t CSVF_BAD_NUMBER = 1u, CSVF_FIELD_COUNT = 2u, CSVF_QUOTE = 4u, CSVF_NULL_IN_REQUIRED = 8u, CSVF_FLOAT_PRECISION = 16u;
void launch_csv_count_lines(hipStream_t s, const uint8_t* text, i64 n, i64 chunk, int nblocks, uint32_t* counts);
void launch_csv_line_starts(hipStream_t s, const uint8_t* text, i64 n, i64 chunk, int nblocks, const uint32_t* block_offsets, i64* starts);
void launch_csv_parse(hipStream_t s, const uint8_t* text, i64 n_bytes, const i64* starts, i64 row0, i64 n_rows, const CsvSpec& S, const CsvOut& O, uint32_t* flags);
void launch_csv_copy_strings(hipStream_t s, const uint8_t* text, const uint32_t* start, const int32_t* offsets, i64 n, uint8_t* out);
enum PqPhys : int32_t { PQ_BOOL = 0, PQ_I32 = 1, PQ_I64 = 2, PQ_I96 = 3 /* nanoseconds of the day + Julian day -> Timestamp(ns) */, PQ_F32 = 4, PQ_F64 = 5, PQ_BYTE_ARRAY = 6, PQ_FLBA = 7 };
enum PqEnc : int32_t { PQE_PLAIN = 0, PQE_DICT = 2, PQE_RLE = 3 };
struct PqPage { i64 src; int32_t bytes; int32_t n_values; i64 row0; int32_t enc; int32_t def_bytes; int32_t def_v2; int32_t dict; };
struct PqDict { i64 values; i64 str_offsets; int32_t n; int32_t pad; };
struct PqCol { int32_t phys, width; int32_t flba_len, optional; void* data; u64* valid; int32_t* str_len; i64* str_src; };
constexpr uint32_t PQF_MALFORMED = 1u, PQF_UNSUPPORTED = 2u;
void launch_pq_decode(hipStream_t s, const uint8_t* file, i64 file_bytes, const PqPage* pages, int n_pages, const PqCol& C, const PqDict* dicts, const uint8_t* dict_values,
                      const int32_t* dict_str_offsets, uint32_t* scratch, i64 scratch_stride, uint32_t* flags);
void launch_pq_copy_strings(hipStream_t s, const uint8_t* file, const uint8_t* dict_values, const i64* src, const int32_t* offsets, i64 n, uint8_t* out);
void launch_pq_dict_strings(hipStream_t s, const uint8_t* file, i64 src, int32_t bytes, int32_t n, int32_t* offsets, uint8_t* out, uint32_t* flags);
void launch_pq_dict_fixed(hipStream_t s, const uint8_t* file, i64 src, int32_t n, int32_t phys, int32_t flba_len, int32_t width, uint8_t* out);

// JIT redirection: while a JitOverride is alive on this thread, the next launch of the kernel family it
// names goes to the hiprtc-compiled function instead of the AOT template instantiation.
struct JitOverride { void* fn = nullptr; int kernel_id = 0; };
JitOverride& jit_override();
template <class... Args>
inline hipError_t jit_launch(void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
  void* a[] = {(void*)&args...};
  return hipModuleLaunchKernel((hipFunction_t)fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)lds, s, a, nullptr);
}
#endif  // GPUQ_JIT

}  // namespace gpuq
