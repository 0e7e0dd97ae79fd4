// mm_fastq.hip — FASTQ text -> PackedSeq records on the device (round 4).
//
// The reference's loader reads FASTA and FASTQ alike (needletail::parse_fastx_file, bench/src/lib.rs:51-82; the
// format is told apart by the first byte, '>' or '@'); rounds 2-3 packed FASTA only and refused FASTQ.  FASTQ is the
// format of READS: its records go straight into the reads / batch entry points.
//
// Semantics restated here (needletail is not in the tree: PARITY UNPINNED, like the FASTA packer; the restatement the
// tests compare with is oracle.fastq_records): a record is FOUR lines - '@' + name, the sequence, '+' (+ optional
// name), the qualities - lines end with '\n' or "\r\n" ('\r' is dropped wherever it stands), the last line may lack
// its '\n', blank lines after the last record are ignored.  Multi-line sequences are not FASTQ as needletail reads it.
// A record's sequence is line 4r + 1; every sequence byte packs as (c >> 1) & 3 like PackedSeqVec::from_ascii.  No
// validation on the device (needletail errors on a record whose third line does not start with '+' or whose quality
// length differs; here such a text packs whatever its lines 4r + 1 hold).
//
// All records go back to back into ONE 2-bit buffer, record r = bases [rec_base[r], rec_base[r + 1]) - the layout of
// the FASTA packer, i.e. what mm_run_batch_device takes, and with a fixed read length what mm_run_reads_device takes.
//
// TWO passes over the text (16 KB per workgroup = two 8 KB pieces, 32 bytes per thread and piece, both pieces of a
// thread in registers) with one small resolve step between them:
//   K1  per chunk: its '\n' bytes, and - for EACH of the four line phases the chunk may start in (line index mod 4) -
//       its sequence bytes (bytes of lines 4r + 1 that are not '\n' / '\r') and record starts (first bytes of lines
//       4r), four 16-bit fields per 64-bit word.  What a byte is depends on the number of newlines in front of it mod 4
//       only, so the four answers are one analysis and a rotation of its four counts.
//   R   (two tiny kernels, see fastq_groups_kernel) exclusive sum of the newline counts = the phase every chunk starts
//       in, which selects the field; exclusive sums of the selected counts = every chunk's first output base and first
//       record.
//   K2  every thread packs its (at most 32) sequence bytes per piece - one run, consecutive in the output - and ORs up
//       to three dwords into the cleared output; the first byte of a line 4r writes the record's table entries.
// Per 32-byte piece of a thread everything is bit arithmetic on 32-bit masks (round 4, second version; the first one
// walked the bytes one by one in three passes and ran at 0.44 TB/s of text): byte-equality masks by SWAR, the number of
// newlines before every byte mod 4 from two prefix-XORs (low bit: parity of the newlines; high bit: parity of the
// newlines that arrive on an odd count), the 2-bit codes of four bytes by one multiply per dword.  Both kernels are
// VALU-bound (SQ_INSTS_VALU x issue rate = kernel time, profiles/r04_e_FASTQ_stalls_*.txt): 32 bytes per thread instead
// of 16 halved the per-thread part (sums over the workgroup, the phase variants, address arithmetic).
#include "mm_common.h"
#include "mm_launch.h"
#include "mm_text.h"

namespace mm {

namespace {

// What the 32 bytes of a thread are, as 32-bit masks (bit i = byte i), whatever line the first byte lies in:
//   nl     '\n' bytes (inside the text)
//   valid  bytes inside the text that are neither '\n' nor '\r'
//   lo, hi the number of newlines among the bytes in front of byte i, mod 4 (bit 0 / bit 1)
//   first  bytes that start a line ('\n' right in front; byte 0: the thread's `starts_line`)
struct FqPiece {
    uint32_t nl, valid, lo, hi, first;
};
// (rel: the thread's first byte of the piece, from the chunk's first byte; left: text bytes from the chunk's first byte
// on, capped at 2^31; has_cr: wave-uniform, does any thread of the wave hold a '\r' - most texts have none)
__device__ __forceinline__ FqPiece analyse(const Text32 &v, uint32_t rel, uint32_t left, bool starts_line, bool has_cr) {
    FqPiece f;
    const uint32_t nin = rel >= left ? 0u : (left - rel >= 32u ? 32u : left - rel);
    const uint32_t inside = nin >= 32u ? 0xffffffffu : ((1u << nin) - 1u);
    f.nl = eq32(v, 0x0a0a0a0au) & inside;
    const uint32_t cr = has_cr ? eq32(v, 0x0d0d0d0du) : 0u;
    f.valid = inside & ~f.nl & ~cr;
    f.lo = pxor32(f.nl) << 1;
    f.hi = pxor32(f.nl & f.lo) << 1;  // (the count's bit 1 flips where a newline arrives on an odd count)
    f.first = (f.nl << 1) | (starts_line ? 1u : 0u);
    return f;
}
// the bytes in front of which the newline count of the piece is t mod 4
__device__ __forceinline__ uint32_t count_is(const FqPiece &f, uint32_t t) {
    return (f.lo ^ ((t & 1u) ? 0u : 0xffffffffu)) & (f.hi ^ ((t & 2u) ? 0u : 0xffffffffu));
}

// rotate the four 8-bit fields of x up by a fields: field s of the result = field (s - a) & 3 of x
__device__ __forceinline__ uint32_t rot_fields(uint32_t x, uint32_t a) {
    return __builtin_amdgcn_alignbit(x, x, (32u - 8u * a) & 31u);
}
// four 8-bit fields -> four 16-bit fields
__device__ __forceinline__ unsigned long long widen_fields(uint32_t x) {
    return (unsigned long long)(x & 0xffu) | ((unsigned long long)(x & 0xff00u) << 8) | ((unsigned long long)(x & 0xff0000u) << 16) |
           ((unsigned long long)(x & 0xff000000u) << 24);
}

// what both kernels start with: the chunk's text, its masks, the newlines in front of every piece of every thread
struct FqChunk {
    Text32 v[kFqPieces];
    FqPiece f[kFqPieces];
    uint32_t before[kFqPieces];  // newlines of the chunk in front of the thread's piece
    uint32_t total_nl;
};
__device__ __forceinline__ void read_chunk(const uint8_t *text, uint64_t n, uint64_t c0, uint32_t (*s_part)[kFqWaves], FqChunk &c) {
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) c.v[p] = load32(text, n, c0 + (uint64_t)p * kFqPiece, threadIdx.x);
    const uint32_t left = n - c0 > 0x7fffffffull ? 0x7fffffffu : (uint32_t)(n - c0);
    const bool has_cr = __ballot((any_eq32(c.v[0], 0x0d0d0d0du) | any_eq32(c.v[1], 0x0d0d0d0du)) != 0u) != 0ull;
    uint32_t nls[kFqPieces];
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        const uint32_t rel = (uint32_t)p * kFqPiece + threadIdx.x * kFqBytesPerThread;
        c.f[p] = analyse(c.v[p], rel, left, starts_line_of(text, n, c0 + rel, c.v[p]), has_cr);
        nls[p] = (uint32_t)__popc(c.f[p].nl);
    }
    c.total_nl = chunk_exclusive(nls, c.before, s_part);
}

// K1: per chunk the newlines, and sequence bytes / record starts for each of the four phases it may start in
__global__ __launch_bounds__(kFqThreads) void fastq_count_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                                 unsigned long long *__restrict__ nl_count,
                                                                 unsigned long long *__restrict__ seq_by_phase,
                                                                 unsigned long long *__restrict__ rec_by_phase) {
    __shared__ uint32_t s_part[kFqPieces][kFqWaves];
    __shared__ unsigned long long s_red[2][kFqWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    FqChunk c;
    read_chunk(text, n, (uint64_t)blockIdx.x * kFqChunk, s_part, c);
    // Field s (chunk starts in phase s): this thread's byte i is a sequence byte iff s + before + count(i) == 1 mod 4.
    // P[j] = bytes whose own count is j; the answer for s is P[(1 - before - s) & 3]: the counts in the order
    // P[0], P[3], P[2], P[1], rotated up by (1 - before) & 3 fields.  Record starts: the same with 0 for 1.
    uint32_t seq8 = 0, rec8 = 0;
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        uint32_t ps = 0, pr = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t m = count_is(c.f[p], j) & c.f[p].valid;
            const uint32_t field = 8u * ((4u - j) & 3u);
            ps |= (uint32_t)__popc(m) << field;
            pr |= (uint32_t)__popc(m & c.f[p].first) << field;
        }
        seq8 += rot_fields(ps, (1u - c.before[p]) & 3u);  // (a field stays below 2 x 32 = 64)
        rec8 += rot_fields(pr, (0u - c.before[p]) & 3u);
    }
    unsigned long long a = widen_fields(seq8), b = widen_fields(rec8);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, kWave);
        b += __shfl_xor(b, d, kWave);
    }
    if (lane == 0) {
        s_red[0][wave] = a;
        s_red[1][wave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long ta = 0, tb = 0;
#pragma unroll
        for (int w = 0; w < kFqWaves; ++w) {
            ta += s_red[0][w];
            tb += s_red[1][w];
        }
        nl_count[blockIdx.x] = c.total_nl;
        seq_by_phase[blockIdx.x] = ta;  // (a field is at most 16 384: no carry between fields)
        rec_by_phase[blockIdx.x] = tb;
    }
}

// R: the chunk functions resolved in two small steps (a first version - ONE workgroup walking all 65 536 chunks of a
// 1 GiB text, sixteen per thread - took 0.28 ms of 1.06: uncoalesced, and two dependent rounds of loads per round).
//   R1  one workgroup per GROUP of 256 chunks, one chunk per thread: the newlines in front of every chunk within its
//       group, and - for each of the four phases the GROUP may start in - the sequence bytes / record starts in front
//       of every chunk within the group (four 32-bit fields in two words) and of the whole group: the chunk's fields
//       rotated down by its newline count, then plain workgroup sums.
//   R2  one workgroup, one group per thread: the line every group starts in -> its phase -> its field -> exclusive sums.
// K2 adds its group's resolved start and its own within-group prefix of the group's phase.
constexpr uint32_t kFqGroup = 256;  // chunks per group = threads of R1
struct FqScratch {
    unsigned long long *nl;        // [chunks]      K1: '\n' bytes of the chunk; R1: newlines in front of it within its group
    unsigned long long *seq, *rec; // [chunks]      K1: four 16-bit fields (the chunk starts in phase s)
    unsigned long long *pre_seq, *pre_rec;  // [2 * chunks]  R1: four 32-bit fields: in front of the chunk within its group
    unsigned long long *g_nl;      // [groups + 1]  R1: '\n' bytes of the group; R2: the line it starts in
    unsigned long long *g_seq, *g_rec;      // [2 * groups]  R1: four 32-bit fields: the whole group
    unsigned long long *g_seq0, *g_rec0;    // [groups + 1]  R2: first output base / first record of the group ([groups]: totals)
};
// four 16-bit fields, field s of the result = field (s + by) & 3 of x, widened to four 32-bit fields (lo: 0, 1; hi: 2, 3)
__device__ __forceinline__ void rotate_widen(unsigned long long x, uint32_t by, unsigned long long *lo, unsigned long long *hi) {
    const uint32_t sh = 16u * (by & 3u);
    const unsigned long long y = sh ? (x >> sh) | (x << (64u - sh)) : x;
    *lo = (y & 0xffffull) | ((y & 0xffff0000ull) << 16);
    *hi = ((y >> 32) & 0xffffull) | (((y >> 32) & 0xffff0000ull) << 16);
}
__device__ __forceinline__ unsigned long long field32(unsigned long long lo, unsigned long long hi, uint32_t s) {
    const unsigned long long w = (s & 2u) ? hi : lo;
    return (s & 1u) ? (w >> 32) : (w & 0xffffffffull);
}
__global__ __launch_bounds__(kFqGroup) void fastq_groups_kernel(FqScratch sc, uint64_t chunks) {
    __shared__ unsigned long long s_wave[kFqWaves];
    const uint64_t c = (uint64_t)blockIdx.x * kFqGroup + threadIdx.x;
    const bool in = c < chunks;
    const unsigned long long nl = in ? sc.nl[c] : 0ull;
    unsigned long long tot_nl;
    const unsigned long long before = block_exclusive64(nl, s_wave, &tot_nl);
    unsigned long long sl, sh, rl, rh;
    rotate_widen(in ? sc.seq[c] : 0ull, (uint32_t)before, &sl, &sh);
    rotate_widen(in ? sc.rec[c] : 0ull, (uint32_t)before, &rl, &rh);
    unsigned long long t0, t1, t2, t3;
    const unsigned long long psl = block_exclusive64(sl, s_wave, &t0), psh = block_exclusive64(sh, s_wave, &t1);
    const unsigned long long prl = block_exclusive64(rl, s_wave, &t2), prh = block_exclusive64(rh, s_wave, &t3);
    if (in) {
        sc.nl[c] = before;
        sc.pre_seq[2 * c] = psl;
        sc.pre_seq[2 * c + 1] = psh;
        sc.pre_rec[2 * c] = prl;
        sc.pre_rec[2 * c + 1] = prh;
    }
    if (threadIdx.x == 0) {
        sc.g_nl[blockIdx.x] = tot_nl;
        sc.g_seq[2 * blockIdx.x] = t0;
        sc.g_seq[2 * blockIdx.x + 1] = t1;
        sc.g_rec[2 * blockIdx.x] = t2;
        sc.g_rec[2 * blockIdx.x + 1] = t3;
    }
}
__global__ __launch_bounds__(kFqGroup) void fastq_resolve_kernel(FqScratch sc, uint64_t groups) {
    __shared__ unsigned long long s_wave[kFqWaves];
    unsigned long long carry_nl = 0, carry_seq = 0, carry_rec = 0;
    for (uint64_t g0 = 0; g0 < groups; g0 += kFqGroup) {  // (one round up to 1 GiB of text)
        const uint64_t g = g0 + threadIdx.x;
        const bool in = g < groups;
        const unsigned long long nl = in ? sc.g_nl[g] : 0ull;
        unsigned long long tot;
        const unsigned long long line = carry_nl + block_exclusive64(nl, s_wave, &tot);
        carry_nl += tot;
        const uint32_t ph = (uint32_t)(line & 3ull);
        const unsigned long long s = in ? field32(sc.g_seq[2 * g], sc.g_seq[2 * g + 1], ph) : 0ull;
        const unsigned long long r = in ? field32(sc.g_rec[2 * g], sc.g_rec[2 * g + 1], ph) : 0ull;
        const unsigned long long s0 = carry_seq + block_exclusive64(s, s_wave, &tot);
        carry_seq += tot;
        const unsigned long long r0 = carry_rec + block_exclusive64(r, s_wave, &tot);
        carry_rec += tot;
        if (in) {
            sc.g_nl[g] = line;
            sc.g_seq0[g] = s0;
            sc.g_rec0[g] = r0;
        }
    }
    if (threadIdx.x == 0) {
        sc.g_nl[groups] = carry_nl;
        sc.g_seq0[groups] = carry_seq;
        sc.g_rec0[groups] = carry_rec;
    }
}

// K2: pack
__global__ __launch_bounds__(kFqThreads) void fastq_pack_kernel(const uint8_t *__restrict__ text, uint64_t n, const FqScratch sc,
                                                                uint32_t *__restrict__ out32, uint64_t out_dwords,
                                                                unsigned long long *__restrict__ rec_base,
                                                                unsigned long long *__restrict__ rec_pos,
                                                                uint64_t max_records, uint64_t pos_bias) {
    __shared__ uint32_t s_part[kFqPieces][kFqWaves];
    // The chunk's output is assembled in LDS (at most 16 384 bases = 1024 dwords, + 1 for its bit offset, + 2 for the
    // last thread's three-dword OR) and leaves as whole dwords; only the first and the last dword, which the chunk
    // shares with its neighbours, are OR-ed into the cleared output.  (Every thread OR-ing its bits straight into
    // global memory, three atomics per thread and piece, took the same time - the kernel is VALU-bound - but wrote
    // 473 MB per GiB of text where this writes 190.)
    constexpr uint32_t kOutDwords = kFqChunk / 16u + 4u;
    __shared__ uint32_t s_out[kOutDwords];
    for (uint32_t i = threadIdx.x; i < kOutDwords; i += kFqThreads) s_out[i] = 0u;  // (ordered by read_chunk's barriers)
    const uint64_t c0 = (uint64_t)blockIdx.x * kFqChunk;
    const uint64_t grp = blockIdx.x / kFqGroup;
    const uint32_t gphase = (uint32_t)(sc.g_nl[grp] & 3ull);  // phase the chunk's group starts in
    const uint32_t phase0 = (gphase + (uint32_t)sc.nl[blockIdx.x]) & 3u;
    const unsigned long long seq0 = sc.g_seq0[grp] + field32(sc.pre_seq[2ull * blockIdx.x], sc.pre_seq[2ull * blockIdx.x + 1], gphase);
    const unsigned long long rec0 = sc.g_rec0[grp] + field32(sc.pre_rec[2ull * blockIdx.x], sc.pre_rec[2ull * blockIdx.x + 1], gphase);
    FqChunk c;
    read_chunk(text, n, c0, s_part, c);
    uint32_t seq_mask[kFqPieces], start_mask[kFqPieces], both[kFqPieces], both_before[kFqPieces];
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        const uint32_t ph = (phase0 + c.before[p]) & 3u;  // phase of the thread's first byte
        seq_mask[p] = count_is(c.f[p], (1u - ph) & 3u) & c.f[p].valid;
        start_mask[p] = count_is(c.f[p], (0u - ph) & 3u) & c.f[p].valid & c.f[p].first;
        both[p] = (uint32_t)__popc(seq_mask[p]) | ((uint32_t)__popc(start_mask[p]) << 16);  // (sums stay below 2^16)
    }
    const uint32_t chunk_seq = chunk_exclusive(both, both_before, s_part) & 0xffffu;  // sequence bytes of the chunk
    const uint64_t q0 = seq0 >> 4;  // the chunk's first output dword
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        const unsigned long long o0 = seq0 + (both_before[p] & 0xffffu);  // first output base of the thread's piece
        const unsigned long long r0 = rec0 + (both_before[p] >> 16);
        const uint32_t sm = seq_mask[p];
        // the thread's sequence bytes are consecutive in the output: at most 64 bits over up to three dwords
        if (sm) {
            // 2-bit codes of the 32 bytes, byte i at bits 2i (one multiply per dword gathers four codes)
            auto codes8 = [](uint32_t x) { return (((x >> 1) & 0x03030303u) * 0x01041040u) >> 24; };
            const Text32 &v = c.v[p];
            const uint32_t clo = codes8(v.d[0]) | (codes8(v.d[1]) << 8) | (codes8(v.d[2]) << 16) | (codes8(v.d[3]) << 24);
            const uint32_t chi = codes8(v.d[4]) | (codes8(v.d[5]) << 8) | (codes8(v.d[6]) << 16) | (codes8(v.d[7]) << 24);
            const unsigned long long codes = (unsigned long long)clo | ((unsigned long long)chi << 32);
            const uint32_t first = (uint32_t)__builtin_ctz(sm), run = sm >> first;
            unsigned long long bits;
            if ((run & (run + 1u)) == 0u) {  // ONE run of bytes (the rule: a piece of one sequence line)
                const uint32_t len = (uint32_t)__popc(sm);
                bits = (codes >> (2u * first)) & (len >= 32u ? ~0ull : ((1ull << (2u * len)) - 1ull));
            } else {  // ('\r' inside a line, or reads shorter than a piece)
                bits = 0;
                uint32_t k = 0;
#pragma unroll
                for (int i = 0; i < (int)kFqBytesPerThread; ++i)
                    if ((sm >> i) & 1u) {
                        bits |= ((codes >> (2 * i)) & 3ull) << (2u * k);
                        ++k;
                    }
            }
            const uint32_t q = (uint32_t)((o0 >> 4) - q0);
            const uint32_t sh = 2u * (uint32_t)(o0 & 15ull);
            const unsigned long long lo = bits << sh;                              // (64 bits shifted by at most 30:
            const uint32_t top = sh ? (uint32_t)(bits >> (64u - sh)) : 0u;         //  96 bits over three dwords)
            if ((uint32_t)lo) atomicOr(&s_out[q], (uint32_t)lo);
            if ((uint32_t)(lo >> 32)) atomicOr(&s_out[q + 1], (uint32_t)(lo >> 32));
            if (top) atomicOr(&s_out[q + 2], top);
        }
        // record table: a record's first base is the number of sequence bytes in front of its '@'
        uint32_t st = start_mask[p], k = 0;
        while (st) {
            const uint32_t i = (uint32_t)__builtin_ctz(st);
            st &= st - 1u;
            const unsigned long long r = r0 + k++;
            if (r < max_records) {
                rec_base[r] = o0 + (uint32_t)__popc(sm & ((1u << i) - 1u));
                if (rec_pos) rec_pos[r] = pos_bias + c0 + (uint64_t)p * kFqPiece + (uint64_t)threadIdx.x * kFqBytesPerThread + i;
            }
        }
    }
    __syncthreads();
    const uint32_t nd = chunk_seq ? (uint32_t)(((seq0 & 15ull) + chunk_seq + 15ull) >> 4) : 0u;  // dwords the chunk touches
    for (uint32_t i = threadIdx.x; i < nd; i += kFqThreads) {
        const uint32_t w = s_out[i];
        const uint64_t q = q0 + i;
        if (w == 0u || q >= out_dwords) continue;  // (the output was cleared)
        if (i == 0 || i + 1 == nd) atomicOr(&out32[q], w);
        else out32[q] = w;
    }
}

__global__ void fastq_finish_kernel(const FqScratch sc, uint64_t groups, unsigned long long *rec_base, uint64_t max_records,
                                    unsigned long long *counts) {
    const unsigned long long bases = sc.g_seq0[groups], recs = sc.g_rec0[groups];
    counts[0] = bases;
    counts[1] = recs;
    if (recs <= max_records) rec_base[recs] = bases;
}

}  // namespace

static uint64_t fastq_chunks(uint64_t n_bytes) { return (n_bytes + kFqChunk - 1) / kFqChunk; }
static uint64_t fastq_groups(uint64_t chunks) { return (chunks + kFqGroup - 1) / kFqGroup; }

uint64_t fastq_scratch_bytes(uint64_t n_bytes) {
    const uint64_t chunks = fastq_chunks(n_bytes), groups = fastq_groups(chunks);
    return (7 * chunks + 7 * (groups + 1)) * sizeof(unsigned long long);
}

int launch_fastq_pack(const uint8_t *d_text, uint64_t n_bytes, uint8_t *d_packed, uint64_t packed_capacity_bytes,
                      unsigned long long *d_rec_base, unsigned long long *d_rec_pos, uint64_t max_records,
                      unsigned long long *d_counts, void *scratch, hipStream_t stream, uint64_t pos_bias) {
    // (pos_bias: added to every record's text position - the caller passed the text from its first '@' on)
    const uint64_t chunks = fastq_chunks(n_bytes), groups = fastq_groups(chunks);
    if (chunks == 0 || chunks >= (1ull << 31)) return -1;
    unsigned long long *q = static_cast<unsigned long long *>(scratch);
    FqScratch sc;
    sc.nl = q, q += chunks;
    sc.seq = q, q += chunks;
    sc.rec = q, q += chunks;
    sc.pre_seq = q, q += 2 * chunks;
    sc.pre_rec = q, q += 2 * chunks;
    sc.g_nl = q, q += groups + 1;
    sc.g_seq = q, q += 2 * (groups + 1);
    sc.g_rec = q, q += 2 * (groups + 1);
    sc.g_seq0 = q, q += groups + 1;
    sc.g_rec0 = q, q += groups + 1;
    const uint64_t out_dwords = packed_capacity_bytes / 4;
    if (out_dwords && hipMemsetAsync(d_packed, 0, out_dwords * 4, stream) != hipSuccess) return -1;
    hipLaunchKernelGGL(fastq_count_kernel, dim3((uint32_t)chunks), dim3(kFqThreads), 0, stream, d_text, n_bytes, sc.nl, sc.seq,
                       sc.rec);
    hipLaunchKernelGGL(fastq_groups_kernel, dim3((uint32_t)groups), dim3(kFqGroup), 0, stream, sc, chunks);
    hipLaunchKernelGGL(fastq_resolve_kernel, dim3(1), dim3(kFqGroup), 0, stream, sc, groups);
    hipLaunchKernelGGL(fastq_pack_kernel, dim3((uint32_t)chunks), dim3(kFqThreads), 0, stream, d_text, n_bytes, sc,
                       reinterpret_cast<uint32_t *>(d_packed), out_dwords, d_rec_base, d_rec_pos, max_records, pos_bias);
    hipLaunchKernelGGL(fastq_finish_kernel, dim3(1), dim3(1), 0, stream, sc, groups, d_rec_base, max_records, d_counts);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace mm
