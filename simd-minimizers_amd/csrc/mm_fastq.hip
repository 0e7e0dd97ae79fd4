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
// Three plain passes over the text (16 KB per workgroup in four rounds of 4 KB, 16 bytes per thread and round):
//   K1  '\n' bytes per chunk                      -> S1 exclusive sum = index of the line every chunk starts in
//   K2  with that index: sequence bytes (bytes of lines 4r + 1 that are not '\n' / '\r') and record starts (first bytes
//       of lines 4r) per chunk                    -> S2 exclusive sums
//   K3  every thread packs its (at most 16, consecutive in the output) sequence bytes and ORs one or two dwords into
//       the cleared output; the first byte of a line 4r writes the record's table entries.
// Not tuned like the FASTA packer's one-pass kernel: ~2.5 passes over the text and atomics for the output.
#include "mm_common.h"
#include "mm_launch.h"

namespace mm {

namespace {

constexpr uint32_t kFqThreads = 256;
constexpr uint32_t kFqBytesPerThread = 16;
constexpr uint32_t kFqPiece = kFqThreads * kFqBytesPerThread;   // 4 KB of text per round of a workgroup
constexpr uint32_t kFqPieces = 4;                               // rounds per workgroup
constexpr uint32_t kFqChunk = kFqPiece * kFqPieces;             // 16 KB of text per workgroup (one entry of the scans)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the thread's 16 text bytes (zeros past the end of the text), from a bounds-checked view of the chunk
__device__ __forceinline__ u32x4 load16(const uint8_t *text, uint64_t n, uint64_t c0, uint32_t t) {
    const uint64_t left = n - c0;
    const uint32_t here = left < kFqPiece ? (uint32_t)left : kFqPiece;
    // (the text pointer may have any alignment: the view starts at the dword that holds byte c0)
    const uintptr_t a = reinterpret_cast<uintptr_t>(text + c0);
    const uint32_t sh = (uint32_t)(a & 3u);
    const __amdgpu_buffer_rsrc_t r =
        // (whole dwords: the bounds check drops a dword that is only partly inside, and the text's last dword may be)
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint32_t *>(a - sh), 0, (int)((here + sh + 3u) & ~3u), 0x00020000);
    const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, t * 16u, 0, 0);
    if (sh == 0) return lo;
    const uint32_t nx = __builtin_amdgcn_raw_buffer_load_b32(r, t * 16u + 16u, 0, 0);
    u32x4 v;
    v.x = __builtin_amdgcn_alignbyte(lo.y, lo.x, sh);
    v.y = __builtin_amdgcn_alignbyte(lo.z, lo.y, sh);
    v.z = __builtin_amdgcn_alignbyte(lo.w, lo.z, sh);
    v.w = __builtin_amdgcn_alignbyte(nx, lo.w, sh);
    return v;
}
__device__ __forceinline__ uint32_t byte_of(const u32x4 &v, int i) {
    const uint32_t w = i < 4 ? v.x : i < 8 ? v.y : i < 12 ? v.z : v.w;
    return (w >> (8 * (i & 3))) & 0xffu;
}

// inclusive sum over the workgroup (256 threads); returns this thread's inclusive value, *total = the workgroup's sum
__device__ __forceinline__ uint32_t block_inclusive(uint32_t v, uint32_t *s_wave, uint32_t *total) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t incl = wave_inclusive_sum(v);
    if (lane == kWave - 1) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < (int)(kFqThreads / kWave); ++w) {
        const uint32_t x = s_wave[w];
        if (w < wave) before += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return incl + before;
}

__global__ __launch_bounds__(kFqThreads) void fastq_newlines_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                                    unsigned long long *__restrict__ nl_count) {
    __shared__ uint32_t s_wave[kFqThreads / kWave];
    uint32_t cnt = 0;
    for (uint32_t p = 0; p < kFqPieces; ++p) {
        const uint64_t c0 = (uint64_t)blockIdx.x * kFqChunk + (uint64_t)p * kFqPiece;
        if (c0 >= n) break;
        const u32x4 v = load16(text, n, c0, threadIdx.x);
        const uint64_t b0 = c0 + (uint64_t)threadIdx.x * kFqBytesPerThread;
#pragma unroll
        for (int i = 0; i < (int)kFqBytesPerThread; ++i)
            if (b0 + i < n && byte_of(v, i) == '\n') ++cnt;
    }
    uint32_t total;
    (void)block_inclusive(cnt, s_wave, &total);
    if (threadIdx.x == 0) nl_count[blockIdx.x] = total;
}

// exclusive sums of one or two arrays of `m` 64-bit counts, in place, by ONE workgroup (m = text bytes / 4096:
// 262 144 for 1 GiB); element m receives the grand total.  Sixteen consecutive elements per thread and round (a first
// version with one element per thread took 1.5 ms per scan: a thousand rounds of three barriers).
constexpr int kFqScanPerThread = 16;
__global__ __launch_bounds__(kFqThreads) void fastq_scan_kernel(unsigned long long *a, unsigned long long *b, uint64_t m) {
    __shared__ unsigned long long s_wave[2][kFqThreads / kWave];
    __shared__ unsigned long long s_carry[2];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (threadIdx.x < 2) s_carry[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i0 = 0; i0 < m; i0 += (uint64_t)kFqThreads * kFqScanPerThread) {
        const uint64_t t0 = i0 + (uint64_t)threadIdx.x * kFqScanPerThread;
        unsigned long long v[2][kFqScanPerThread], sum[2] = {0ull, 0ull}, incl[2];
#pragma unroll
        for (int u = 0; u < kFqScanPerThread; ++u) {
            v[0][u] = t0 + u < m ? a[t0 + u] : 0ull;
            v[1][u] = (b && t0 + u < m) ? b[t0 + u] : 0ull;
            sum[0] += v[0][u];
            sum[1] += v[1][u];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            unsigned long long x = sum[q];
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const unsigned long long y = __shfl_up(x, d, kWave);
                if (lane >= d) x += y;
            }
            incl[q] = x;
            if (lane == kWave - 1) s_wave[q][wave] = x;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q == 1 && !b) continue;
            unsigned long long run = s_carry[q] + incl[q] - sum[q];
            for (int w = 0; w < wave; ++w) run += s_wave[q][w];
#pragma unroll
            for (int u = 0; u < kFqScanPerThread; ++u) {
                if (t0 + u < m) (q == 0 ? a : b)[t0 + u] = run;
                run += v[q][u];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned long long tot = 0;
                for (int w = 0; w < (int)(kFqThreads / kWave); ++w) tot += s_wave[q][w];
                s_carry[q] += tot;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a[m] = s_carry[0];
        if (b) b[m] = s_carry[1];
    }
}

// What a thread knows about its 16 bytes once the line index of its first byte is known.
struct FqThread {
    uint32_t seq_mask;    // bit i: byte i is a sequence byte (line 4r + 1, not '\n' / '\r')
    uint32_t start_mask;  // bit i: byte i is the first byte of a line 4r (a record's '@')
};
__device__ __forceinline__ FqThread classify(const u32x4 &v, uint64_t b0, uint64_t n, unsigned long long line0,
                                             bool first_is_line_start) {
    FqThread r{0u, 0u};
    unsigned long long line = line0;
    bool at_start = first_is_line_start;
    // (bytes past the end of the text read as 0: neither '\n' nor counted, see `inside`)
    uint32_t ph = (uint32_t)(line & 3ull);
#pragma unroll
    for (int i = 0; i < (int)kFqBytesPerThread; ++i) {
        const bool inside = b0 + i < n;
        const uint32_t c = byte_of(v, i);
        const bool nl = inside && c == '\n';
        const bool other = inside && !nl;
        if (other && at_start && ph == 0u && c != '\r') r.start_mask |= 1u << i;  // (a blank line 4r starts no record)
        if (other && ph == 1u && c != '\r') r.seq_mask |= 1u << i;
        ph = nl ? ((ph + 1u) & 3u) : ph;
        at_start = nl ? true : (other ? false : at_start);
    }
    (void)line;
    return r;
}

// line index of the thread's first byte and whether that byte starts a line: from the chunk's line index, the
// newlines of the threads before it in the chunk, and the byte in front of it
__device__ __forceinline__ void thread_context(const uint8_t *text, uint64_t n, uint64_t c0, const u32x4 &v,
                                               unsigned long long chunk_line0, uint32_t *s_wave,
                                               unsigned long long *line0, bool *starts_line, uint32_t *piece_newlines) {
    const uint64_t b0 = c0 + (uint64_t)threadIdx.x * kFqBytesPerThread;
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < (int)kFqBytesPerThread; ++i)
        if (b0 + i < n && byte_of(v, i) == '\n') ++cnt;
    uint32_t total;
    const uint32_t incl = block_inclusive(cnt, s_wave, &total);
    *piece_newlines = total;
    *line0 = chunk_line0 + (incl - cnt);
    *starts_line = b0 == 0 || (b0 < n && text[b0 - 1] == '\n');
}

__global__ __launch_bounds__(kFqThreads) void fastq_count_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                                 const unsigned long long *__restrict__ line_base,
                                                                 unsigned long long *__restrict__ seq_count,
                                                                 unsigned long long *__restrict__ rec_count) {
    __shared__ uint32_t s_wave[kFqThreads / kWave];
    unsigned long long lines = line_base[blockIdx.x];
    uint32_t nseq = 0, nrec = 0;
    for (uint32_t p = 0; p < kFqPieces; ++p) {
        const uint64_t c0 = (uint64_t)blockIdx.x * kFqChunk + (uint64_t)p * kFqPiece;
        if (c0 >= n) break;
        const u32x4 v = load16(text, n, c0, threadIdx.x);
        unsigned long long line0;
        bool sl;
        uint32_t nl;
        thread_context(text, n, c0, v, lines, s_wave, &line0, &sl, &nl);
        lines += nl;
        const FqThread f = classify(v, c0 + (uint64_t)threadIdx.x * kFqBytesPerThread, n, line0, sl);
        nseq += __popc(f.seq_mask);
        nrec += __popc(f.start_mask);
    }
    uint32_t tot_seq, tot_rec;
    (void)block_inclusive(nseq, s_wave, &tot_seq);
    (void)block_inclusive(nrec, s_wave, &tot_rec);
    if (threadIdx.x == 0) {
        seq_count[blockIdx.x] = tot_seq;
        rec_count[blockIdx.x] = tot_rec;
    }
}

__global__ __launch_bounds__(kFqThreads) void fastq_pack_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                                const unsigned long long *__restrict__ line_base,
                                                                const unsigned long long *__restrict__ seq_base,
                                                                const unsigned long long *__restrict__ rec_base_chunk,
                                                                uint32_t *__restrict__ out32, uint64_t out_dwords,
                                                                unsigned long long *__restrict__ rec_base,
                                                                unsigned long long *__restrict__ rec_pos,
                                                                uint64_t max_records) {
    __shared__ uint32_t s_wave[kFqThreads / kWave];
    unsigned long long lines = line_base[blockIdx.x], seq_run = seq_base[blockIdx.x], rec_run = rec_base_chunk[blockIdx.x];
    for (uint32_t p = 0; p < kFqPieces; ++p) {
    const uint64_t c0 = (uint64_t)blockIdx.x * kFqChunk + (uint64_t)p * kFqPiece;
    if (c0 >= n) break;
    const uint64_t b0 = c0 + (uint64_t)threadIdx.x * kFqBytesPerThread;
    const u32x4 v = load16(text, n, c0, threadIdx.x);
    unsigned long long line0;
    bool sl;
    uint32_t nl;
    thread_context(text, n, c0, v, lines, s_wave, &line0, &sl, &nl);
    lines += nl;
    const FqThread f = classify(v, b0, n, line0, sl);
    uint32_t tot_s, tot_r;
    const uint32_t nseq = __popc(f.seq_mask), nrec = __popc(f.start_mask);
    const unsigned long long o0 = seq_run + (block_inclusive(nseq, s_wave, &tot_s) - nseq);   // first output base
    const unsigned long long r0 = rec_run + (block_inclusive(nrec, s_wave, &tot_r) - nrec);
    seq_run += tot_s;
    rec_run += tot_r;
    // the thread's sequence bytes are consecutive in the output: at most 32 bits over one or two dwords
    if (nseq) {
        unsigned long long bits = 0;
        uint32_t k = 0;
#pragma unroll
        for (int i = 0; i < (int)kFqBytesPerThread; ++i)
            if ((f.seq_mask >> i) & 1u) {
                bits |= (unsigned long long)((byte_of(v, i) >> 1) & 3u) << (2u * k);
                ++k;
            }
        const uint64_t q = o0 >> 4;
        const uint32_t sh = 2u * (uint32_t)(o0 & 15ull);
        const unsigned long long wide = bits << sh;  // (32 bits shifted by at most 30: fits 64)
        if (q < out_dwords && (uint32_t)wide) atomicOr(&out32[q], (uint32_t)wide);
        if (q + 1 < out_dwords && (uint32_t)(wide >> 32)) atomicOr(&out32[q + 1], (uint32_t)(wide >> 32));
    }
    // record table: a record's first base is the number of sequence bytes in front of its '@'
    if (nrec) {
        uint32_t seq_before = 0, k = 0;
#pragma unroll
        for (int i = 0; i < (int)kFqBytesPerThread; ++i) {
            if ((f.start_mask >> i) & 1u) {
                const unsigned long long r = r0 + k;
                if (r < max_records) {
                    rec_base[r] = o0 + seq_before;
                    if (rec_pos) rec_pos[r] = b0 + i;
                }
                ++k;
            }
            if ((f.seq_mask >> i) & 1u) ++seq_before;
        }
    }
    }  // pieces
}

__global__ void fastq_finish_kernel(const unsigned long long *seq_base, const unsigned long long *rec_base_chunk, uint64_t chunks,
                                    unsigned long long *rec_base, uint64_t max_records, unsigned long long *counts) {
    const unsigned long long bases = seq_base[chunks], recs = rec_base_chunk[chunks];
    counts[0] = bases;
    counts[1] = recs;
    if (recs <= max_records) rec_base[recs] = bases;
}

}  // namespace

uint64_t fastq_scratch_bytes(uint64_t n_bytes) {
    const uint64_t chunks = (n_bytes + kFqChunk - 1) / kFqChunk;
    return 3 * (chunks + 1) * sizeof(unsigned long long);
}

int launch_fastq_pack(const uint8_t *d_text, uint64_t n_bytes, uint8_t *d_packed, uint64_t packed_capacity_bytes,
                      unsigned long long *d_rec_base, unsigned long long *d_rec_pos, uint64_t max_records,
                      unsigned long long *d_counts, void *scratch, hipStream_t stream) {
    const uint64_t chunks = (n_bytes + kFqChunk - 1) / kFqChunk;
    if (chunks == 0 || chunks >= (1ull << 31)) return -1;
    unsigned long long *line_base = static_cast<unsigned long long *>(scratch);
    unsigned long long *seq_base = line_base + (chunks + 1), *rec_chunk = seq_base + (chunks + 1);
    const uint64_t out_dwords = packed_capacity_bytes / 4;
    if (out_dwords && hipMemsetAsync(d_packed, 0, out_dwords * 4, stream) != hipSuccess) return -1;
    hipLaunchKernelGGL(fastq_newlines_kernel, dim3((uint32_t)chunks), dim3(kFqThreads), 0, stream, d_text, n_bytes, line_base);
    hipLaunchKernelGGL(fastq_scan_kernel, dim3(1), dim3(kFqThreads), 0, stream, line_base, (unsigned long long *)nullptr, chunks);
    hipLaunchKernelGGL(fastq_count_kernel, dim3((uint32_t)chunks), dim3(kFqThreads), 0, stream, d_text, n_bytes, line_base,
                       seq_base, rec_chunk);
    hipLaunchKernelGGL(fastq_scan_kernel, dim3(1), dim3(kFqThreads), 0, stream, seq_base, rec_chunk, chunks);
    hipLaunchKernelGGL(fastq_pack_kernel, dim3((uint32_t)chunks), dim3(kFqThreads), 0, stream, d_text, n_bytes, line_base,
                       seq_base, rec_chunk, reinterpret_cast<uint32_t *>(d_packed), out_dwords, d_rec_base, d_rec_pos,
                       max_records);
    hipLaunchKernelGGL(fastq_finish_kernel, dim3(1), dim3(1), 0, stream, seq_base, rec_chunk, chunks, d_rec_base, max_records,
                       d_counts);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace mm
