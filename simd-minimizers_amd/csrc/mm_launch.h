// mm_launch.h — host-side launch interface between the C ABI (mm_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "mm_common.h"

namespace mm {

struct RunArgs {
    SeqView seq;
    HashTables ht;
    uint32_t k, w;
    int canonical_windows;
    uint32_t mode;  // 0 minimizers, 1 closed syncmers, 2 open syncmers
    uint64_t win_begin, win_end;
    OutParams out;
    // skip-ambiguous windows: bit i set = window i holds an ambiguous base (null = plain PackedSeq)
    const uint32_t *wamb;
    uint32_t wamb_dwords;
    // batch mode of the fused family: many sequences, one launch (device tables; null = one sequence)
    const BatchSeq *batch_seqs;
    const BatchTile *batch_tile_seq;
    unsigned long long *batch_offsets;  // n + 1 entries
    uint32_t batch_n;
    uint64_t batch_tiles;
    // fused path
    uint32_t nblk;  // w-blocks per lane (0 = default)
    uint64_t work_windows;  // windows of the whole run (0 = unknown): short runs get shorter lanes (more tiles)
    int use_ticket; // tile ids from an atomic ticket instead of blockIdx.x (safe mode)
    uint64_t status_avail = 0;  // 8-byte words allocated behind out.status (0 = not known to the launcher)
    // Epoch tag of this launch's look-back status words (fused family; kEpochShift in mm_common.h).  Non-zero: the
    // caller guarantees that no word behind out.status carries this tag yet, and the launcher clears nothing.
    // 0: the launcher clears the words it uses (the protocol of rounds 1-3).
    uint32_t status_epoch = 0;
    bool append = false;        // *out.total holds the outputs before this launch (else the run starts at 0)
    // generic path
    void *scratch;
    uint64_t generic_round_windows;
    // optional HIP events recorded right around the dominant kernel
    hipEvent_t timing_start, timing_stop;
};

// Longest lane of the fused family in windows (16-bit element positions inside a lane; legal_nblk, the reads-mode
// launcher and the whole-rounds tuner all keep to it).
constexpr uint32_t kFusedMaxLaneWindows = 60000u;
// Bytes behind the last base of a run's last window that the run may still TOUCH (never use): a lane that starts inside
// the window range walks its whole length with the windows past the range masked, and its sequence loads run two load
// groups ahead.  The launcher's own bound - what mm_device_group_upload_range keeps resident behind an entry's share and
// what mm_run_sharded_device's residency check allows for (round 4 carried a literal 24 576 there).
uint64_t fused_overread_bytes();

// ---- fused family (mm_fused_*.hip): one kernel, specialised per w
bool fused_supported(uint32_t k, uint32_t w, int canonical_windows, int hasher_canonical);
uint64_t fused_status_words(const RunArgs &a);
// 8-byte words reserved per tile status (the look-back words of consecutive tiles are spaced apart)
uint64_t fused_status_stride();
// windows per tile for this plan / output flavour (what a batch's tile table is built from)
uint32_t fused_tile_windows(const RunArgs &a);
// blocks per lane that make a batch fill whole rounds of resident workgroups (0 = keep the default);
// the caller puts it into RunArgs::nblk before building the tile table
uint32_t fused_batch_nblk(const RunArgs &a, const uint64_t *n_windows, uint64_t n_seqs);
// The tile table of a batch launch: whole tiles per sequence in input order (a sequence's last tile may be partial),
// the last round of the launch tapered.  Appends to `tiles`; returns false if the table would pass 2^31 tiles.
// launch plan of a single-sequence run as the kernel would get it (MM_TAPER_SLOTS set: no device needed); see mm_fused.hip
int fused_debug_plan(const RunArgs &a, unsigned long long *out /* [7] */);
void fused_debug_lds(const RunArgs &a, unsigned long long *out2);  // {bytes of the lane lists, bytes of the skip-ambiguous landing area}
// *nblk_out receives the longest lane of the table (RunArgs::nblk of the launch: it sizes the lists).
bool fused_batch_tiles(const RunArgs &a, const uint64_t *n_windows, uint64_t n_seqs, std::vector<BatchTile> &tiles,
                       uint32_t *nblk_out);
// returns 0, -1 (HIP failure) or -2 (no kernel for this plan: take the generic family;
// fused_unavailable_reason() says why)
int launch_fused(const RunArgs &a, hipStream_t stream);
const char *fused_unavailable_reason();
// window sizes with a prebuilt instance (sequence mode / reads mode), ascending; returns their number
int fused_prebuilt_windows(bool canonical, bool reads, uint32_t *out, int capacity);

// ---- split path of the fused family (walk_kernel in mm_fused_impl.h + mm_split.hip): the walk dumps its lists
// and exits, persistent expander workgroups on a second stream turn them into positions
struct SplitBuffers {
    uint8_t *dump = nullptr;                     // tiles x dump slot
    uint64_t dump_bytes = 0;
    unsigned long long *tile_status = nullptr;   // one word per tile
    uint64_t status_words = 0;
    void *redo_list = nullptr;                   // one 16-byte entry per tile
    uint64_t redo_entries = 0;
    uint32_t *redo_n = nullptr;
    unsigned long long *carry = nullptr;         // outputs before the run (append mode)
    hipStream_t aux = nullptr;                   // the expander's stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};
// what a run needs (tiles of the run and bytes of its dump area); tiles == 0: the plan has no split path
void split_requirements(const RunArgs &a, uint64_t *tiles, uint64_t *dump_bytes);
// default policy + MM_SPLIT (0 / 1) override: whether this run should take the split path
bool split_wanted(const RunArgs &a);
// returns 0, -1 (HIP failure) or -2 (no walk kernel for this plan: take the fused kernel)
int launch_split(const RunArgs &a, const SplitBuffers &b, hipStream_t stream);

// ---- reads mode of the fused family: a batch of short reads at a fixed stride, one lane per read
struct ReadsArgs {
    SeqView seq;
    HashTables ht;
    uint32_t k, w;
    uint32_t mode;  // 0 minimizers (prebuilt instances), 1 / 2 closed / open syncmers (run-time specialised)
    int canonical_windows;
    uint64_t n_reads;
    uint32_t read_stride, read_len;
    const uint32_t *read_lens;           // device, optional
    const unsigned long long *read_starts = nullptr;  // device, optional: reads back to back (n_reads + 1 entries)
    unsigned long long *read_offsets;    // device, n_reads + 1
    const uint32_t *wamb;                // window ambiguity bits over the whole buffer span (or null)
    uint32_t wamb_dwords;
    OutParams out;
    int use_ticket;
    uint32_t status_epoch = 0;  // as in RunArgs
    hipEvent_t timing_start, timing_stop;
};
bool fused_reads_supported(uint32_t w, int canonical_windows, int hasher_canonical, uint32_t mode = 0);
uint64_t fused_reads_status_words(const ReadsArgs &a);
// returns 0, -1 (HIP failure), -2 (no instance), -3 (reads too long for the LDS lists)
int launch_fused_reads(const ReadsArgs &a, hipStream_t stream);

// ---- lane-table launches of the reads-mode kernels (round 6): reads / sequences of ANY lengths in one launch at full
// lane occupancy.  mm_lanes.hip builds the table (LaneSeg, mm_common.h) on the device from the reads' starts and lengths.
struct SegSource {  // where read r starts (bases from the first base of the span) and how long it is; device arrays
    const unsigned long long *starts;  // null: r * stride
    const uint32_t *lens;              // null: starts[r + 1] - starts[r] when `starts` is given, max_len otherwise
    unsigned long long stride;
    uint32_t max_len;                  // longer reads are cut to it
};
struct SegPlan {
    uint32_t nblk, S, list_cap, lds_bytes;  // lane length (blocks, windows), entries and bytes of the lane lists
    uint64_t lanes_cap, tiles;              // upper bound of the lanes (a multiple of 256) and the grid
};
struct SegBuffers {
    LaneSeg *table;         // plan.tiles * 256 entries
    uint32_t *tile_origin;  // plan.tiles
    uint32_t *seg_first;    // n_reads + 1
    uint32_t *blk_sums;     // lane_table_blocks(n_reads) + 1
};
uint64_t lane_table_blocks(uint64_t n_reads);
// (`error`: the run's error words, OutParams::error - code 5 when the reads need more lanes than plan.lanes_cap)
int launch_lane_table(const SegSource &src, uint64_t n_reads, uint32_t l, const SegPlan &plan, const SegBuffers &b,
                      uint32_t *error, hipStream_t stream);
// lane length and grid of a lane-table launch over `n_reads` reads of `total_bases` bases in all (nblk_want: blocks per
// lane, 0 = the default lanes of the plan).  Returns 0, -2 (no kernel) or -3 (no lane length fits).
int fused_segments_plan(const ReadsArgs &a, uint64_t total_bases, uint32_t nblk_want, SegPlan *plan);
// the table's kernels and the walk, on `stream`.  Returns 0, -1 (HIP failure) or -2 (no kernel for this plan).
int launch_fused_segments(const ReadsArgs &a, const SegSource &src, const SegPlan &plan, const SegBuffers &b,
                          hipStream_t stream);

// ---- run-time specialisation (mm_jit.hip): window sizes without a prebuilt instance
constexpr uint32_t kJitMaxW = 128;  // ring registers: 256 VGPRs + AGPRs still hold W = 128 without scratch
bool jit_enabled();                 // MM_JIT=0 switches it off (then such w take the generic family)
// walk = true: mm::walk_kernel (the split path's walk) instead of mm::fused_kernel
hipFunction_t jit_fused_kernel(uint32_t w, bool canon, bool hash_rc, int mode, bool sk, bool reads,
                               std::string *err, bool walk = false);

// ---- generic family (mm_generic.hip): any k / w
uint64_t generic_scratch_bytes(uint64_t round_windows, uint32_t w);
uint64_t generic_status_words(uint64_t round_windows);
int launch_generic(const RunArgs &a, hipStream_t stream);

// ---- auxiliary kernels (mm_aux.hip)
int launch_values_u64(SeqView seq, uint32_t len, int canonical, const uint32_t *d_pos,
                      uint64_t n_pos, unsigned long long *d_values, hipStream_t stream);
int launch_values_u128(SeqView seq, uint32_t len, int canonical, const uint32_t *d_pos,
                       uint64_t n_pos, unsigned long long *d_values, hipStream_t stream);
int launch_pack_ascii(const uint8_t *d_ascii, uint64_t n, uint8_t *d_packed, hipStream_t stream);
int launch_pack_ascii_n(const uint8_t *d_ascii, uint64_t n, uint8_t *d_packed, uint8_t *d_amb,
                        hipStream_t stream);
// bit i of d_out = window i (bases [i, i+l)) holds an ambiguous base; windows [win_begin-1, win_end)
int launch_window_ambiguity(const uint32_t *d_amb, uint32_t amb_dwords, uint64_t bit0, uint32_t l,
                            uint64_t win_begin, uint64_t win_end, uint32_t *d_out, hipStream_t stream);
// ---- FASTA text -> packed records (mm_fasta.hip)
uint64_t fasta_scratch_bytes(uint64_t n_bytes);
int launch_fasta_pack(const uint8_t *d_text, uint64_t n_bytes, uint8_t *d_packed, uint64_t packed_capacity_bytes,
                      unsigned long long *d_rec_base, unsigned long long *d_rec_pos, uint64_t max_records,
                      unsigned long long *d_counts, void *scratch, hipStream_t stream, bool one_pass = false,
                      uint32_t *d_error = nullptr);
// ---- the same in two passes of mask arithmetic (mm_fasta2.hip; the default since late round 4)
uint64_t fasta2_scratch_bytes(uint64_t n_bytes);
int launch_fasta_pack2(const uint8_t *d_text, uint64_t n_bytes, uint8_t *d_packed, uint64_t packed_capacity_bytes,
                       unsigned long long *d_rec_base, unsigned long long *d_rec_pos, uint64_t max_records,
                       unsigned long long *d_counts, void *scratch, hipStream_t stream);

// ---- FASTQ text -> packed records (mm_fastq.hip): four-line records, the sequences of lines 4r + 1
uint64_t fastq_scratch_bytes(uint64_t n_bytes);
int launch_fastq_pack(const uint8_t *d_text, uint64_t n_bytes, uint8_t *d_packed, uint64_t packed_capacity_bytes,
                      unsigned long long *d_rec_base, unsigned long long *d_rec_pos, uint64_t max_records,
                      unsigned long long *d_counts, void *scratch, hipStream_t stream, uint64_t pos_bias = 0);
// copies to / from page-locked host memory by a kernel (mm_aux.hip; the host entry point's alternative mechanisms)
int launch_copy_range(const uint32_t *d_src, uint32_t *dst_host_alias, const unsigned long long *d_range, uint64_t cap,
                      uint32_t workgroups, hipStream_t stream);
int launch_copy16(const void *src, void *dst, uint64_t n16, uint32_t workgroups, hipStream_t stream);
// diagnostics: shader clock while other kernels run (out: 2 words per workgroup)
int launch_clock_probe(unsigned long long *d_out, uint32_t workgroups, uint64_t ticks, hipStream_t stream);
int launch_generate(uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *d_packed,
                    hipStream_t stream);

}  // namespace mm
