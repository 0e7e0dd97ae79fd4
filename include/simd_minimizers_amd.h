/*
 * simd_minimizers_amd.h — C ABI of the MI355X (gfx950) minimizer engine.
 *
 * This is the drop-in boundary for the hot path of rust-seq/simd-minimizers
 * v3.0.0 (packed-seq decode -> ntHash -> sliding-window min -> dedup/collect,
 * the canonical variant, the syncmer filter, super-k-mer indices and k-mer
 * values).  The reference has no FFI of its own: the path sits behind its
 * Rust builder API (src/lib.rs:225-577).  Each entry point below names the
 * reference item it replaces; INTEGRATION.md shows the Rust `extern "C"`
 * binding a maintainer would add.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function
 * returns 0 (MM_OK) or a negative MM_ERR_* code and never aborts.  The HIP
 * kernels are the only compute path: there is no CPU fallback.
 *
 * Current device: an entry point selects its workspace's device (a device
 * group's entries one after the other) while it runs and RESTORES the calling
 * thread's current device before it returns - a caller that works with several
 * GPUs finds hipGetDevice() unchanged after every call (round 6).  Device
 * pointers passed in must belong to the device the call works on: a
 * workspace's device, the root entry's device for the gather calls.
 */
#ifndef SIMD_MINIMIZERS_AMD_H
#define SIMD_MINIMIZERS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ types */

/* seq-hash KmerHasher as data (call sites src/minimizers.rs:24,44,61,85,143; src/lib.rs:391).
 *   h_fw(i) = fw_xor ^ XOR_j rotl(fw[s[i+j]], rot*(k-1-j));  h_rc(i) = rc_xor ^ XOR_j rotl(rc[s[i+j]], rot*j)
 *   h = canonical ? h_fw + h_rc (wrapping) : h_fw
 * The tables cross the ABI as data, so a seeded hasher (`new_with_seed`, src/lib.rs:157) or another
 * hasher of this rolling rot-xor form is a parameter and not a rebuild.  fw_xor / rc_xor are constants
 * XORed onto the strand hashes (0 for NtHasher; folded into the kernels' tables, no cost per base);
 * `kind` says which seq-hash type the tables stand for (informational: the kernels only see the
 * tables). */
typedef struct mm_hasher {
    uint32_t fw[4];
    uint32_t rc[4];
    uint32_t rot;
    uint32_t canonical;
    uint32_t fw_xor;
    uint32_t rc_xor;
    uint32_t kind; /* mm_hasher_kind_t */
} mm_hasher_t;

typedef enum mm_hasher_kind { MM_HASHER_NT = 0, MM_HASHER_MUL = 1, MM_HASHER_ANTILEX = 2 } mm_hasher_kind_t;

/* Builder<_, _, _, SYNCMER> (src/lib.rs:221-225): 0 minimizers, 1 closed, 2 open syncmers */
typedef enum mm_mode { MM_MINIMIZERS = 0, MM_CLOSED_SYNCMERS = 1, MM_OPEN_SYNCMERS = 2 } mm_mode_t;

/* Which kernel family a run used (diagnostics; all are HIP kernels).  MM_PATH_SPLIT is the fused family's
 * two-stream form: the same walk, its lists expanded to positions by a second, concurrent kernel. */
typedef enum mm_path { MM_PATH_FUSED = 1, MM_PATH_GENERIC = 2, MM_PATH_SPLIT = 3 } mm_path_t;

typedef struct mm_device_group mm_device_group_t; /* one workspace per listed device (several-device calls) */
typedef struct mm_plan mm_plan_t;           /* immutable (k, w, hasher, mode): the Builder     */
typedef struct mm_workspace mm_workspace_t; /* per-stream device scratch: the thread-local CACHE
                                               of src/lib.rs:217-219, src/collect.rs:124-126   */

/* One entry per assert!/panic! on the reference path. */
enum {
    MM_OK = 0,
    MM_ERR_W_ZERO = -1,               /* src/sliding_min.rs:91,227 */
    MM_ERR_W_TOO_LARGE = -2,          /* src/sliding_min.rs:92-95,228 */
    MM_ERR_LEN_TOO_LARGE = -3,        /* src/sliding_min.rs:96-99,229 */
    MM_ERR_EVEN_L = -4,               /* src/canonical.rs:13-16,43-46 */
    MM_ERR_HASHER_NOT_CANONICAL = -5, /* src/minimizers.rs:81,139 */
    MM_ERR_OPEN_EVEN_W = -6,          /* src/syncmers.rs:24-29 */
    MM_ERR_K_ZERO = -7,
    MM_ERR_CAPACITY = -8,             /* caller's output buffer too small; *out_count holds the need */
    MM_ERR_BAD_MODE = -9,             /* src/lib.rs:437; super-k-mers with syncmers, src/lib.rs:339 */
    MM_ERR_NULL = -10,
    MM_ERR_VALUE_LEN = -11,           /* values_u64 needs len <= 32, values_u128 len <= 64 */
    MM_ERR_FORMAT = -12,              /* mm_fasta_pack_device: the text is FASTQ ('@' first), not FASTA */
    MM_ERR_NO_DEVICE = -20,           /* no HIP device: the engine has no CPU fallback */
    MM_ERR_HIP = -21,                 /* a HIP call failed; see mm_last_error() */
    MM_ERR_ALLOC = -22,
    MM_ERR_ORDER = -23                /* mm_workspace_check: a look-back of an asynchronous run timed out */
};

const char *mm_strerror(int code);
/* Text of the last HIP failure seen by this thread ("" if none). */
const char *mm_last_error(void);
/* Number of visible HIP devices (0 if none; never initialises a device context). */
int mm_device_count(void);

/* ----------------------------------------------------------------- hasher */

/* NtHasher::<CANONICAL>::new(k) (seq-hash 0.2.0; src/lib.rs:391). */
int mm_default_hasher(mm_hasher_t *out, int canonical);
/* MulHasher::<CANONICAL>::new(k) and AntiLexHasher::<CANONICAL>::new(k) (seq-hash 0.2.0; src/lib.rs:71-72,
 * exercised by src/test.rs:81-83,107-109).  PARITY UNPINNED: their arithmetic is not in the reference
 * tree and the reference holds no known-answer vector for them, so these fill the tables with this
 * engine's restatement of the published idea - mulHash: the character value times a pseudo-random
 * constant in NtHasher's rolling form; anti-lex: the k-mer's own base-4 value with the first base
 * inverted - and a caller who has the real crate puts ITS per-base values into mm_hasher_t instead.
 * Everything downstream (windows, ties, strand vote, collectors) is the pinned path. */
int mm_mul_hasher(mm_hasher_t *out, int canonical);
int mm_antilex_hasher(mm_hasher_t *out, uint32_t k, int canonical);

/* ------------------------------------------------------------------- plan */

/* minimizers / canonical_minimizers / closed_syncmers / canonical_closed_syncmers /
 * open_syncmers / canonical_open_syncmers (src/lib.rs:240-321) + .hasher() (:327).
 * `canonical_windows` selects the strand-vote tie-break (src/minimizers.rs:74-166);
 * `hasher == NULL` means H::new(k) with the matching CANONICAL (src/lib.rs:391-394).
 * Validates every precondition the reference asserts and returns the matching error. */
int mm_plan_create(mm_plan_t **out, uint32_t k, uint32_t w, int canonical_windows, mm_mode_t mode,
                   const mm_hasher_t *hasher);
void mm_plan_destroy(mm_plan_t *plan);
/* len of the values: k for minimizers, k+w-1 for syncmers (src/lib.rs:439-447) */
uint32_t mm_plan_value_len(const mm_plan_t *plan);

/* -------------------------------------------------------------- workspace */

/* Device scratch bound to one HIP device and one stream.  `hip_stream` may be NULL
 * (a private stream is created; it is a blocking stream, i.e. ordered against the legacy
 * default stream) or a hipStream_t owned by the caller. */
int mm_workspace_create(mm_workspace_t **out, int device, void *hip_stream);
void mm_workspace_destroy(mm_workspace_t *ws);
int mm_workspace_sync(mm_workspace_t *ws);
/* Status of the ASYNCHRONOUS runs issued on this workspace since the last check (the reference has
 * no counterpart: its calls are synchronous; this is the completion half of the *_async entry
 * points).  Waits for the workspace stream, then returns MM_OK, or
 *   MM_ERR_ORDER  the fused kernel's look-back timed out in one of them (workgroups were not
 *                 dispatched in index order): that run's output and count are invalid; every later
 *                 run on this workspace takes its tile ids from an atomic ticket, so repeating the
 *                 runs issued since the last check gives the right result;
 *   MM_ERR_HIP    a kernel refused to run (mm_last_error() says why).
 * The synchronous entry points check (and repeat the run) themselves. */
int mm_workspace_check(mm_workspace_t *ws);
/* Force the generic (any k, any w) kernel family instead of the fused one (testing). */
int mm_workspace_force_generic(mm_workspace_t *ws, int on);
/* Windows per lane of the fused kernel, in units of w (0 = built-in default). Tuning knob. */
int mm_workspace_set_blocks_per_lane(mm_workspace_t *ws, uint32_t nblk);
/* HIP-event timing of the dominant kernel: when enabled every launch of the hot kernel is
 * bracketed by events on the workspace stream; read back the sum and the launch count. */
int mm_workspace_enable_timing(mm_workspace_t *ws, int on);
int mm_workspace_kernel_time(mm_workspace_t *ws, double *total_ms, uint64_t *launches,
                             int reset);
/* Family used by the last run (mm_path_t). */
int mm_workspace_last_path(const mm_workspace_t *ws);
/* 1 when the last reads / batch run on this workspace was a LANE-TABLE launch (round 6): one launch of the reads-mode
 * kernel whose lanes are segments of the reads - a read longer than a lane takes consecutive lanes - so reads and
 * sequences of any lengths (Builder::run per read / contig, src/lib.rs:378; the reference's `short` experiment spans
 * lengths 16 .. 16 384, bench/src/bin/paper.rs:62-115) fill every tile.  mm_run_reads_device*, mm_run_packed_reads_*
 * take it when the longest read exceeds a default lane, mm_run_batch_device for batches of short sequences that lie in
 * ONE device allocation within 2^32 bases of one another (one descriptor then covers them all; sequences in separate
 * allocations keep tiles of their own, whose loads are clamped to each sequence's bytes); diagnostics only, results are
 * the same on every path. */
int mm_workspace_last_lane_table(const mm_workspace_t *ws);
/* Window sizes w for which the library carries a PREBUILT fused kernel (every other w <= 128 is specialised at
 * first use, larger ones take the generic family): canonical_windows 0 / 1 selects the forward / canonical
 * instances, reads_mode 0 / 1 the sequence-mode / reads-mode ones.  Writes up to `capacity` sizes in ascending
 * order to `out` (may be null) and returns how many there are.  No GPU needed.  The test-suite takes its list of
 * instances to compare with the oracle from here, so that none can ship untested. */
int mm_prebuilt_window_sizes(int canonical_windows, int reads_mode, uint32_t *out, int capacity);
/* Bytes behind the last base of a run's last window that a launch of the fused family may still TOUCH (never use: a lane
 * that starts inside the window range walks its whole length with the windows past the range masked, and its loads run
 * ahead).  The launcher's own bound: mm_device_group_upload_range keeps that much resident behind every entry's share and
 * mm_run_sharded_device's residency check allows for it.  No GPU needed. */
uint64_t mm_fused_overread_bytes(void);

/* -------------------------------------------------------------------- run */

/* Builder::run on a device-resident PackedSeq (src/lib.rs:378-448, :545-576).
 *
 *  d_packed      device pointer to 2-bit bases, 4 per byte, base i at bits 2(i%4) of byte i/4
 *                (packed-seq PackedSeq), codes A0 C1 T2 G3; any byte alignment
 *  packed_bytes  readable bytes at d_packed (>= ceil((base_offset + n_bases)/4))
 *  base_offset   index of the sequence's first base inside the buffer (PackedSeq slices may
 *                start inside a byte: src/test.rs:42-45)
 *  win_begin/end half-open range of WINDOW indices to produce (0 .. n_bases-l+1); the range
 *                form lets one long sequence be sharded across GPUs with absolute positions and
 *                an exact dedup at the seam (src/collect.rs:265-271).  win_end = UINT64_MAX
 *                means "to the last window".
 *  d_out_pos     device buffer for positions (minimizer modes) / window indices (syncmer modes)
 *  d_out_sk      optional device buffer for super-k-mer start indices (.super_kmers(), :341)
 *  capacity      elements available in d_out_pos (and d_out_sk)
 *  d_count       optional device uint64 receiving the number of outputs
 *
 * Asynchronous on the workspace stream.  Output order equals window order.  Completion status:
 * mm_workspace_check(). */
int mm_run_device_async(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                        uint64_t packed_bytes, uint64_t base_offset, uint64_t n_bases,
                        uint64_t win_begin, uint64_t win_end, uint32_t *d_out_pos,
                        uint32_t *d_out_sk, uint64_t capacity, uint64_t *d_count);

/* Same, then waits and returns the count; MM_ERR_CAPACITY if it exceeded `capacity`. */
int mm_run_device(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                  uint64_t packed_bytes, uint64_t base_offset, uint64_t n_bases,
                  uint64_t win_begin, uint64_t win_end, uint32_t *d_out_pos, uint32_t *d_out_sk,
                  uint64_t capacity, uint64_t *out_count);

/* Builder::run on host memory: H2D copy, kernels, D2H copy (what a Rust caller holding a
 * PackedSeqVec / Vec<u32> binds).  `out_pos == NULL` only counts. */
int mm_run_host(const mm_plan_t *plan, mm_workspace_t *ws, const uint8_t *packed,
                uint64_t base_offset, uint64_t n_bases, uint32_t *out_pos, uint32_t *out_sk,
                uint64_t capacity, uint64_t *out_count);

/* AsciiSeq input (src/lib.rs:92-98): packs (c >> 1) & 3 on the device, then runs. */
int mm_run_host_ascii(const mm_plan_t *plan, mm_workspace_t *ws, const uint8_t *ascii,
                      uint64_t n_bases, uint32_t *out_pos, uint32_t *out_sk, uint64_t capacity,
                      uint64_t *out_count);

/* ----------------------------------------------------------------- values */

/* Output::values_u64 (src/lib.rs:584-612): k-mer (minimizers) or l-mer (syncmers) at each
 * position, min(fwd, revcomp) when `canonical`.  Device-resident positions and values. */
int mm_values_u64_device_async(mm_workspace_t *ws, const void *d_packed, uint64_t packed_bytes,
                               uint64_t base_offset, uint64_t n_bases, uint32_t len,
                               int canonical, const uint32_t *d_pos, uint64_t n_pos,
                               uint64_t *d_values);
int mm_values_u64_host(mm_workspace_t *ws, const uint8_t *packed, uint64_t base_offset,
                       uint64_t n_bases, uint32_t len, int canonical, const uint32_t *pos,
                       uint64_t n_pos, uint64_t *values);

/* Output::values_u128 (src/lib.rs:587-629): len <= 64; value i is stored little-endian as
 * values[2i] (low 64 bits) and values[2i+1] (high 64 bits). */
int mm_values_u128_device_async(mm_workspace_t *ws, const void *d_packed, uint64_t packed_bytes,
                                uint64_t base_offset, uint64_t n_bases, uint32_t len,
                                int canonical, const uint32_t *d_pos, uint64_t n_pos,
                                uint64_t *d_values);
int mm_values_u128_host(mm_workspace_t *ws, const uint8_t *packed, uint64_t base_offset,
                        uint64_t n_bases, uint32_t len, int canonical, const uint32_t *pos,
                        uint64_t n_pos, uint64_t *values);

/* Page-locked host memory for the host entry points.  Any host pointer works; with buffers from
 * mm_host_alloc the copies to and from the device run in both directions at once (97 GB/s aggregate
 * against 56 GB/s for pageable memory on the round-1 box), which the pipelined long-sequence path
 * of mm_run_host exploits. */
int mm_host_alloc(void **out, uint64_t bytes);
void mm_host_free(void *p);

/* ------------------------------------------------------------------ batch */

/* Many independent sequences (contigs) with one plan: what the reference does by calling
 * Builder::run once per sequence (bench/src/bin/paper.rs:410-431).  Sequence s lives at
 * d_packed[s] (device pointers, host array).  Positions are sequence-local and are written back
 * to back into d_out_pos; out_offsets[s] .. out_offsets[s+1] (host array of n_seqs+1 entries)
 * delimit sequence s.  With a fused kernel for the plan the whole batch is ONE launch (every tile
 * looks its sequence up in a device table), so thousands of contigs cost no more launches than one
 * chromosome; other plans take one launch per sequence on the workspace stream. */
int mm_run_batch_device(const mm_plan_t *plan, mm_workspace_t *ws, uint64_t n_seqs,
                        const void *const *d_packed, const uint64_t *packed_bytes,
                        const uint64_t *base_offsets, const uint64_t *n_bases,
                        uint32_t *d_out_pos, uint32_t *d_out_sk, uint64_t capacity,
                        uint64_t *out_offsets);

/* Batched short reads (the read-mapping / k-mer-counting shape: millions of 100-300 bp reads, each
 * an independent Builder::run, src/lib.rs:378): read r is the bases
 * [base_offset + r * read_stride, + len_r) of one packed device buffer, len_r = d_read_lens[r]
 * (<= read_len) or read_len when d_read_lens is NULL.  Positions are read-local and written back
 * to back; d_out_offsets (device, n_reads + 1 entries) delimits the reads.  Plans run as ONE launch
 * with one lane per read (prebuilt kernels for minimizers at the common window sizes, kernels
 * specialised at first use for syncmers, super-k-mer indices and other w <= 128); only reads too
 * long for a lane's LDS list or w > 128 take one launch per read - same results. */
int mm_run_reads_device_async(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                              uint64_t packed_bytes, uint64_t base_offset, uint64_t n_reads,
                              uint32_t read_stride, uint32_t read_len, const uint32_t *d_read_lens,
                              uint32_t *d_out_pos, uint64_t capacity, uint64_t *d_out_offsets,
                              uint64_t *d_count);
int mm_run_reads_device(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                        uint64_t packed_bytes, uint64_t base_offset, uint64_t n_reads,
                        uint32_t read_stride, uint32_t read_len, const uint32_t *d_read_lens,
                        uint32_t *d_out_pos, uint64_t capacity, uint64_t *d_out_offsets,
                        uint64_t *out_count);

/* --------------------------------------------------------------- PackedNSeq */

/* Builder::run_skip_ambiguous_windows (src/lib.rs:451-496) on canonical_minimizers /
 * canonical_closed_syncmers / canonical_open_syncmers plans
 * (canonical_minimizers_skip_ambiguous_windows, src/minimizers.rs:169-214): a window with an
 * ambiguous base among its l = k+w-1 bases yields SKIPPED = u32::MAX-1 (src/minimizers.rs:18),
 * which the collectors then drop (collect_and_dedup_into::<true>, src/collect.rs:128-285 with
 * SKIP_MAX; src/syncmers.rs:113-120,154-164).  A PackedNSeq crosses the ABI as two arrays: the
 * PackedSeq bytes (2-bit codes; an ambiguous character carries the lossy code (c>>1)&3) and the
 * ambiguity bits, base i at bit (amb_offset + i) % 8 of byte (amb_offset + i) / 8 (packed-seq 5.0.0
 * BitSeq, not in the reference tree: layout inferred).  Non-canonical plans return
 * MM_ERR_HASHER_NOT_CANONICAL (assert src/minimizers.rs:176); there is no super-k-mer flavour. */
int mm_run_skip_ambiguous_device_async(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                                       uint64_t packed_bytes, uint64_t base_offset, const void *d_amb,
                                       uint64_t amb_bytes, uint64_t amb_offset, uint64_t n_bases,
                                       uint64_t win_begin, uint64_t win_end, uint32_t *d_out_pos,
                                       uint64_t capacity, uint64_t *d_count);
int mm_run_skip_ambiguous_device(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                                 uint64_t packed_bytes, uint64_t base_offset, const void *d_amb,
                                 uint64_t amb_bytes, uint64_t amb_offset, uint64_t n_bases,
                                 uint64_t win_begin, uint64_t win_end, uint32_t *d_out_pos,
                                 uint64_t capacity, uint64_t *out_count);
int mm_run_skip_ambiguous_host(const mm_plan_t *plan, mm_workspace_t *ws, const uint8_t *packed,
                               uint64_t base_offset, const uint8_t *amb, uint64_t amb_offset,
                               uint64_t n_bases, uint32_t *out_pos, uint64_t capacity,
                               uint64_t *out_count);
/* PackedNSeqVec::from_ascii (call site src/test.rs:436) then the run, from a host ASCII buffer. */
int mm_run_skip_ambiguous_host_ascii(const mm_plan_t *plan, mm_workspace_t *ws, const uint8_t *ascii,
                                     uint64_t n_bases, uint32_t *out_pos, uint64_t capacity,
                                     uint64_t *out_count);
/* The same with super-k-mer indices (Builder::super_kmers + run per read, src/lib.rs:341,545-576):
 * d_out_sk[j] = read-local index of the first window that selected d_out_pos[j].  Minimizer plans
 * only (MM_ERR_BAD_MODE otherwise, like src/lib.rs:339). */
int mm_run_reads_superkmers_device_async(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                                         uint64_t packed_bytes, uint64_t base_offset, uint64_t n_reads,
                                         uint32_t read_stride, uint32_t read_len,
                                         const uint32_t *d_read_lens, uint32_t *d_out_pos,
                                         uint32_t *d_out_sk, uint64_t capacity, uint64_t *d_out_offsets,
                                         uint64_t *d_count);
int mm_run_reads_superkmers_device(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                                   uint64_t packed_bytes, uint64_t base_offset, uint64_t n_reads,
                                   uint32_t read_stride, uint32_t read_len, const uint32_t *d_read_lens,
                                   uint32_t *d_out_pos, uint32_t *d_out_sk, uint64_t capacity,
                                   uint64_t *d_out_offsets, uint64_t *out_count);

/* Batched short reads with ambiguity bits (same layout rules as mm_run_reads_device; read r's
 * ambiguity bits start at bit amb_offset + r * read_stride). */
int mm_run_reads_skip_ambiguous_device_async(const mm_plan_t *plan, mm_workspace_t *ws,
                                             const void *d_packed, uint64_t packed_bytes,
                                             uint64_t base_offset, const void *d_amb, uint64_t amb_bytes,
                                             uint64_t amb_offset, uint64_t n_reads, uint32_t read_stride,
                                             uint32_t read_len, const uint32_t *d_read_lens,
                                             uint32_t *d_out_pos, uint64_t capacity,
                                             uint64_t *d_out_offsets, uint64_t *d_count);
int mm_run_reads_skip_ambiguous_device(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                                       uint64_t packed_bytes, uint64_t base_offset, const void *d_amb,
                                       uint64_t amb_bytes, uint64_t amb_offset, uint64_t n_reads,
                                       uint32_t read_stride, uint32_t read_len,
                                       const uint32_t *d_read_lens, uint32_t *d_out_pos,
                                       uint64_t capacity, uint64_t *d_out_offsets, uint64_t *out_count);

/* ------------------------------------------------------------------ input */

/* PackedSeqVec::from_ascii on the device: out byte i/4 |= ((c>>1)&3) << 2(i%4). */
int mm_pack_ascii_device_async(mm_workspace_t *ws, const uint8_t *d_ascii, uint64_t n_bases,
                               uint8_t *d_packed /* ceil(n/4) bytes */);
/* PackedNSeqVec::from_ascii on the device: the packed bytes as above plus one ambiguity bit per
 * base (set for every character that is not ACGT / acgt), d_amb = ceil(n/8) bytes. */
int mm_pack_ascii_n_device_async(mm_workspace_t *ws, const uint8_t *d_ascii, uint64_t n_bases,
                                 uint8_t *d_packed /* ceil(n/4) bytes */, uint8_t *d_amb);
/* FASTA text -> PackedSeq records on the device: what the reference's loader does on the CPU with
 * needletail::parse_fastx_file + PackedSeqVec::from_ascii per record (bench/src/lib.rs:51-82).  A record
 * starts with '>' at the start of a line, its header runs to the end of that line, its sequence is every
 * following line up to the next header with '\n' and '\r' removed (bytes before the first header are
 * ignored); every sequence byte packs as (c >> 1) & 3.  ALL records go back to back into d_packed (4-byte
 * aligned, packed_capacity_bytes a multiple of 4; n_bytes / 4 + 8 always suffices): record r = bases
 * [d_rec_base[r], d_rec_base[r + 1]) of it - pass d_packed + base / 4 with base_offset = base % 4 to
 * mm_run_batch_device.  d_rec_text_pos[r] (optional) = byte offset of the record's '>' in the text (the
 * caller slices the header from there).  d_counts[0] = bases, d_counts[1] = records found; records past
 * max_records are counted but not tabulated.  The text must be shorter than 2^32 bytes.  Two passes over the text
 * (mm_fasta2.hip: every 16 KB chunk's effect on the header / record state as a composable function, then the packing);
 * nothing in them depends on line lengths or on the order workgroups start in, so the call has no failure mode of its
 * own.  (MM_FASTA_KERNEL=lines selects the one-pass kernel of rounds 3-4, which gives up on texts whose lines are
 * shorter than 16 bytes on average and on a look-back time-out - the synchronous call below then repeats the text with
 * the three-pass kernels, an asynchronous caller gets MM_ERR_ORDER from mm_workspace_check(); MM_FASTA_KERNEL=three
 * takes those from the start.  Both are kept as cross-checks.) */
int mm_fasta_pack_device_async(mm_workspace_t *ws, const uint8_t *d_text, uint64_t n_bytes,
                               uint8_t *d_packed, uint64_t packed_capacity_bytes,
                               uint64_t *d_rec_base /* [max_records + 1] */,
                               uint64_t *d_rec_text_pos /* [max_records] or NULL */, uint64_t max_records,
                               uint64_t *d_counts /* [2] */);
/* The same, synchronous: out_counts[0..1] receive the counts; MM_ERR_CAPACITY when the bases did not fit
 * d_packed or the records did not fit the table (the counts say what is needed: out_counts[0] against
 * 4 * packed_capacity_bytes, out_counts[1] against max_records).  A text whose first non-blank byte is '@'
 * is FASTQ, which the reference's loader reads through the same call (needletail::parse_fastx_file): since
 * round 4 this entry point looks at that byte and packs FASTQ with mm_fastq_pack_device_async (same outputs).
 * The asynchronous FASTA entry point above does not look and would pack no record from a FASTQ text. */
int mm_fasta_pack_device(mm_workspace_t *ws, const uint8_t *d_text, uint64_t n_bytes, uint8_t *d_packed,
                         uint64_t packed_capacity_bytes, uint64_t *d_rec_base, uint64_t *d_rec_text_pos,
                         uint64_t max_records, uint64_t *d_counts, uint64_t *out_counts /* [2] */);
/* Reads packed BACK TO BACK (round 4): read r = bases [d_read_starts[r], d_read_starts[r + 1]) of one packed buffer -
 * the layout mm_fastq_pack_device_async / mm_fasta_pack_device write (d_rec_base), so millions of reads of ANY
 * lengths run in ONE launch of the reads-mode kernel (mm_run_batch_device gives every sequence tiles of its own,
 * which is right for contigs and wasteful for reads).  d_read_starts: n_reads + 1 device entries; total_bases =
 * d_read_starts[n_reads] (the packer's count of bases); max_read_len: no read is longer (a longer one is cut to it,
 * like a d_read_lens entry above read_len).  Positions are read-local, d_out_offsets[r] .. [r + 1] delimit read r's;
 * d_out_sk (or NULL): super-k-mer indices.  Builder::run per read, src/lib.rs:378. */
int mm_run_packed_reads_device_async(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                                     uint64_t packed_bytes, uint64_t base_offset, uint64_t n_reads,
                                     const uint64_t *d_read_starts /* [n_reads + 1] */, uint64_t total_bases,
                                     uint32_t max_read_len, uint32_t *d_out_pos, uint32_t *d_out_sk /* or NULL */,
                                     uint64_t capacity, uint64_t *d_out_offsets /* [n_reads + 1] */, uint64_t *d_count);
int mm_run_packed_reads_device(const mm_plan_t *plan, mm_workspace_t *ws, const void *d_packed,
                               uint64_t packed_bytes, uint64_t base_offset, uint64_t n_reads,
                               const uint64_t *d_read_starts, uint64_t total_bases, uint32_t max_read_len,
                               uint32_t *d_out_pos, uint32_t *d_out_sk, uint64_t capacity, uint64_t *d_out_offsets,
                               uint64_t *out_count);
/* The same from HOST memory in one call: n_reads reads packed back to back in `packed` (read r = bases
 * [read_starts[r], read_starts[r + 1]), starts on the host, non-decreasing), one upload, ONE launch, one download.
 * What a caller that looped Builder::run over its reads does instead: a synchronous call costs about 28 us whatever
 * the length, so a per-read loop runs 10-100 x slower than the reference's CPU on 150-base reads, and this call does
 * millions of them at PCIe speed.  out_offsets[r] .. [r + 1] delimit read r's (read-local) positions. */
int mm_run_packed_reads_host(const mm_plan_t *plan, mm_workspace_t *ws, const uint8_t *packed, uint64_t n_reads,
                             const uint64_t *read_starts /* [n_reads + 1] */, uint32_t max_read_len, uint32_t *out_pos,
                             uint32_t *out_sk /* or NULL */, uint64_t capacity, uint64_t *out_offsets /* [n_reads + 1] */,
                             uint64_t *out_count);
/* FASTQ text -> packed records (round 4; mm_fastq.hip): four-line records ('@' name, sequence, '+', qualities;
 * "\r\n" or '\n', a last line without '\n', blank lines after the last record); the sequence of every record is
 * packed like a FASTA record's, same output layout: record r = bases [d_rec_base[r], d_rec_base[r + 1]),
 * d_rec_text_pos[r] = byte offset of its '@', d_counts = {bases, records}.  Reads of one length go straight into
 * mm_run_reads_device (read_stride = read_len), any lengths into mm_run_batch_device.  needletail is not in the
 * reference tree: parity unpinned like the FASTA packer's; no validation of '+' lines or quality lengths.
 * The text has to START with the first record's '@' (a line's role is the number of newlines in front of it mod 4);
 * mm_fasta_pack_device cuts blank bytes in front of it off before it calls this and keeps the text positions absolute. */
int mm_fastq_pack_device_async(mm_workspace_t *ws, const uint8_t *d_text, uint64_t n_bytes,
                               uint8_t *d_packed, uint64_t packed_capacity_bytes,
                               uint64_t *d_rec_base /* [max_records + 1] */,
                               uint64_t *d_rec_text_pos /* [max_records] or NULL */, uint64_t max_records,
                               uint64_t *d_counts /* [2] */);
/* ------------------------------------------------------------------ several devices from one call
 * The reference's parallel driver is host code: rayon over the contigs, one Builder::run each
 * (bench/src/bin/paper.rs:442-459).  A device group holds one workspace (stream, scratch) per listed device; a
 * device may be listed more than once.  Both calls run one host thread per entry, keep the count exchange on
 * the host (nothing crosses between the devices) and deliver ONE dense result in the caller's host buffers. */
int mm_device_group_create(mm_device_group_t **out, const int *devices, int n_devices);
void mm_device_group_destroy(mm_device_group_t *group);
int mm_device_group_size(const mm_device_group_t *group);
/* One sequence cut into n equal window ranges, one per entry of the group: absolute positions, exact seam (a
 * range starts by comparing with the window before it, which is the reference's rule for joining lanes,
 * src/collect.rs:265-271; syncmers have no such rule).  Same result as mm_run_host.  MM_ERR_CAPACITY: *out_count
 * holds the need. */
int mm_run_sharded_host(const mm_plan_t *plan, mm_device_group_t *group, const uint8_t *packed,
                        uint64_t base_offset, uint64_t n_bases, uint32_t *out_pos, uint32_t *out_sk /* or NULL */,
                        uint64_t capacity, uint64_t *out_count);
/* n_seqs independent sequences (contigs) placed greedily, longest first, on the entries of the group; every
 * entry runs its sequences in one batch launch.  Positions are sequence-local and lie in input order:
 * sequence s = out_pos[out_offsets[s] .. out_offsets[s + 1]).  Same result as mm_run_batch_device. */
int mm_run_batch_sharded_host(const mm_plan_t *plan, mm_device_group_t *group, uint64_t n_seqs,
                              const uint8_t *const *packed, const uint64_t *base_offsets /* or NULL */,
                              const uint64_t *n_bases, uint32_t *out_pos, uint32_t *out_sk /* or NULL */,
                              uint64_t capacity, uint64_t *out_offsets /* [n_seqs + 1] */);

/* ---- device-resident shards (round 4).  north_star's multi-GPU shape: the sequence already lives in HBM, every
 * device walks its window range with one asynchronous launch, the positions STAY on their devices, and "at most" a
 * gather moves them to one device over xGMI.  No host buffers, no PCIe transfer, no host thread per device.
 *
 * The sequence: mm_device_group_upload copies the caller's packed bytes to every device of the group ONCE (kept
 * until the next upload or the group's end); mm_device_group_adopt takes device pointers the caller already
 * holds instead - d_packed[i] on the device of entry i, each addressing the same packed_bytes of the same
 * sequence, not owned by the group.  (Every entry addresses the whole sequence so that positions are absolute;
 * an entry only READS the bytes of its own window range and a halo of k + w - 1 bases.) */
int mm_device_group_upload(mm_device_group_t *group, const uint8_t *packed, uint64_t packed_bytes);
/* The same for ONE run shape: every entry receives only the bytes its window range of an N-way split of
 * (base_offset, n_bases) can read - its share plus a halo that covers any k + w - 1 below 2^17 - so the sequence crosses
 * the host link once in total, not once per device (the whole extent is still allocated on every device: offsets stay
 * absolute).  mm_run_sharded_device refuses (MM_ERR_NULL, mm_last_error() names the ranges) a run whose shape needs bytes
 * an entry does not hold.  MM_ERR_CAPACITY: the bases do not fit packed_bytes. */
int mm_device_group_upload_range(mm_device_group_t *group, const uint8_t *packed, uint64_t packed_bytes,
                                 uint64_t base_offset, uint64_t n_bases);
int mm_device_group_adopt(mm_device_group_t *group, const void *const *d_packed /* [size] */, uint64_t packed_bytes);
/* Builder::run over the resident sequence, cut into mm_device_group_size(group) equal window ranges (absolute
 * positions, exact seam: range i starts by comparing with the window before it, src/collect.rs:265-271, so the
 * shards laid end to end ARE the single-device result).  One asynchronous launch per entry, issued from the
 * calling thread; the call returns when all have finished.  The positions (and super-k-mer indices when
 * want_superkmers != 0) are left in result buffers the group owns and grows, one per entry: see
 * mm_device_group_result.  counts[i] (may be NULL) receives entry i's count, *total (may be NULL) their sum. */
int mm_run_sharded_device(const mm_plan_t *plan, mm_device_group_t *group, uint64_t base_offset, uint64_t n_bases,
                          int want_superkmers, uint64_t *counts /* [size] or NULL */, uint64_t *total);
/* Entry i's shard of the last mm_run_sharded_device: device pointers on that entry's device (valid until the next
 * run on the group), its count and its window range.  Any out pointer may be NULL. */
int mm_device_group_result(const mm_device_group_t *group, int entry, uint32_t **d_pos, uint32_t **d_sk,
                           uint64_t *count, uint64_t *win_begin, uint64_t *win_end);
/* The optional exchange: the shards of the last run, dense and in window order, into d_dst_pos (and d_dst_sk) on
 * the device of entry `root` - device-to-device copies (hipMemcpyPeerAsync: over xGMI between the GPUs of a node),
 * all in flight together; no RCCL dependency.  *total receives the number of positions; MM_ERR_CAPACITY when they
 * do not fit `capacity` (nothing is copied then).  The copies run on the SOURCE entries' streams and the call waits for
 * them; they are not ordered against work the caller has queued on the destination: d_dst_pos / d_dst_sk must be idle
 * (no kernel or copy of the caller's still reading or writing them) when this is called.  MM_ERR_NULL when the group's
 * last run was a batch run (mm_run_batch_sharded_device: see mm_device_group_gather_batch) - the two resident modes
 * share the result buffers, each call invalidates the other mode's results. */
int mm_device_group_gather(mm_device_group_t *group, int root, uint32_t *d_dst_pos, uint32_t *d_dst_sk /* or NULL */,
                           uint64_t capacity, uint64_t *total);

/* ---- device-resident BATCHES (round 4): north_star's "sharded by contig across the GPUs of a node, gather of the
 * positions".  mm_device_group_upload_batch places n_seqs independent sequences greedily, longest first, on the
 * entries of the group and copies each to ITS entry's device only (kept until the next batch upload / the group's
 * end).  mm_run_batch_sharded_device runs every entry's sequences in ONE batch launch on its device (Builder::run per
 * sequence, sequence-local positions, bench/src/bin/paper.rs:425-431,442-459); the positions stay on the devices.
 * n_bases[s] (and base_offsets[s], or NULL) describe sequence s of the upload; out_counts[s] (may be NULL) receives
 * its count, *total their sum.  mm_device_group_batch_result hands out where a sequence's positions lie;
 * mm_device_group_gather_batch copies all sequences, in input order, into device memory of entry `root`
 * (device-to-device) and fills the n_seqs + 1 offsets.  Like mm_run_sharded_device the run issues one asynchronous
 * launch per entry from the calling thread and then waits for them in turn (round 5; round 4 ran a host thread per
 * entry); its destination rule for the gather is mm_device_group_gather's. */
int mm_device_group_upload_batch(mm_device_group_t *group, uint64_t n_seqs, const uint8_t *const *packed,
                                 const uint64_t *packed_bytes);
int mm_run_batch_sharded_device(const mm_plan_t *plan, mm_device_group_t *group, const uint64_t *base_offsets /* or NULL */,
                                const uint64_t *n_bases, int want_superkmers, uint64_t *out_counts /* [n_seqs] or NULL */,
                                uint64_t *total);
int mm_device_group_batch_result(const mm_device_group_t *group, uint64_t seq, int *entry, uint32_t **d_pos,
                                 uint32_t **d_sk, uint64_t *count);
int mm_device_group_gather_batch(mm_device_group_t *group, int root, uint32_t *d_dst_pos, uint32_t *d_dst_sk /* or NULL */,
                                 uint64_t capacity, uint64_t *out_offsets /* [n_seqs + 1] */);

/* Diagnostics: the launch plan of the fused kernel as the host lays it out - lane length, tiles, the tapered tail
 * (DESIGN.md 4.1 "Launch geometry").  With MM_TAPER_SLOTS=<workgroup slots> in the environment no device is needed:
 * the CPU test-suite checks with it that the tiles tile every window range exactly.  n_seqs == 0: ONE sequence of
 * n_windows[0] windows, out7 = {blocks per lane, tiles, first tapered tile, tiles per taper level, last level's blocks
 * per lane, first tapered window, windows per block of a tile}.  n_seqs > 0: a batch, the tile table itself (tile t =
 * sequence, first window, blocks per lane; out7[0] = the longest lane); MM_ERR_CAPACITY when it holds more than
 * tile_capacity tiles (*n_tiles says how many).  mode: 0 minimizers, 1 / 2 closed / open syncmers, 3 minimizers with
 * super-k-mer indices (whose 16-bit list entries bound the lanes), 4 minimizers over a PackedNSeq (the skip-ambiguous walk's
 * lane rules).  With MM_TAPER_SLOTS set the single-sequence plan also applies the one-round rule (a run of 0.6 .. 1 round of
 * that many slots gets one tile per slot).
 * mm_debug_launch_lds: the dynamic LDS of the same single-sequence launch, out2 = {bytes of the lane lists, bytes of the
 * skip-ambiguous walk's landing area in front of them (0 without ambiguity bits)} - the CPU suite checks that the two fit
 * the CU as often as the kernel's register bound lets workgroups share it. */
int mm_debug_launch_plan(uint32_t w, int canonical_windows, int mode, uint64_t n_seqs, const uint64_t *n_windows,
                         uint64_t *out7, uint32_t *tile_seq, uint32_t *tile_win0, uint32_t *tile_nblk,
                         uint64_t tile_capacity, uint64_t *n_tiles);
int mm_debug_launch_lds(uint32_t w, int canonical_windows, int mode, uint64_t n_windows, uint64_t *out2);
/* Diagnostics of the lane-table launches (round 6).  mm_debug_lane_plan: the lane length and grid the host
 * chooses for n_reads reads of total_bases bases - out6 = {blocks per lane, windows per lane S, entries per lane list, bytes of
 * the lane lists, upper bound of the lanes (a multiple of 256), tiles}; mode as in mm_debug_launch_plan (3: with super-k-mer
 * indices, whose packed 16-bit entries bound S; 4: over a PackedNSeq); blocks_per_lane 0 = the default.  No device needed.
 * mm_debug_last_lane_table: copies the table the workspace's LAST lane-table run built to the host - lane i =
 * {start, win0, count, read} in out4[4 i .. 4 i + 3] - up to `capacity` lanes; *n_lanes = lanes of the (padded) table.  The
 * GPU test-suite checks with it that the lanes tile every read's windows exactly. */
int mm_debug_lane_plan(uint32_t k, uint32_t w, int canonical_windows, int mode, uint64_t n_reads, uint64_t total_bases,
                       uint32_t blocks_per_lane, uint64_t *out6);
int mm_debug_last_lane_table(mm_workspace_t *ws, uint32_t *out4, uint64_t capacity, uint64_t *n_lanes);
/* Diagnostics: the shader clock while other work runs on the device.  _begin starts a handful of sleeping
 * single-wave workgroups on a stream of the workspace's own that sample the shader cycle counter against the
 * 100 MHz real-time counter for duration_us; _end waits for them and returns the mean clock in GHz (bench.py
 * brackets its timed loop with the pair: the VALU bound of roofline.valu is priced at THIS clock). */
/* Diagnostics: the host link of this box - host -> device alone, device -> host alone and BOTH AT ONCE (copy engines, two
 * streams), `bytes` each way between the caller's (page-locked) buffers and device memory the call allocates.
 * out_GBps[0..2] = the two one-way rates and the SUM of both directions while they run together; the floor of
 * mm_run_host follows from the third, which is less than the sum of the first two (bench.py: end_to_end.link). */
int mm_link_probe(mm_workspace_t *ws, const void *host_in, void *host_out, uint64_t bytes, double *out_GBps /* [3] */);
int mm_clock_probe_begin(mm_workspace_t *ws, uint64_t duration_us);
int mm_clock_probe_end(mm_workspace_t *ws, double *ghz);
/* Deterministic synthetic PackedSeq generator G of BASELINE.md §4, written on the device. */
int mm_generate_device_async(mm_workspace_t *ws, uint64_t seed, uint64_t first_base,
                             uint64_t n_bases, uint8_t *d_packed /* ceil(n/4) bytes */);

#ifdef __cplusplus
}
#endif
#endif
