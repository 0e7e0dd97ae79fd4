/*
 * mm_oracle.h — CPU restatement of the simd-minimizers hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Parity status: the reference (rust-seq/simd-minimizers v3.0.0) is Rust and
 * cannot be built in this image (no cargo/rustc); the hash arithmetic and the
 * PackedSeq layout live in the un-vendored crates seq-hash 0.2.0 and
 * packed-seq 5.0.0.  This restatement is PINNED by every known-answer vector
 * the reference's own tests/doctests hold for the path (tests/golden/
 * reference_vectors.json: src/lib.rs:92-99, :109-129, :132-140,
 * src/test.rs:334-356, :401-415, :484-515, :576-597).  For k=21/31 and seeded
 * hashers the reference has no known-answer test: parity there is against
 * this restatement ("parity unpinned" beyond the k=5 vectors).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef MM_ORACLE_H
#define MM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ntHash-style k-mer hasher (seq-hash 0.2.0 NtHasher<CANONICAL>, not in tree;
 * call sites src/minimizers.rs:24,44,61,85,143; src/lib.rs:391).
 *   h_fw(i) = fw_xor ^ XOR_j rotl(fw[s[i+j]], rot*(k-1-j))
 *   h_rc(i) = rc_xor ^ XOR_j rotl(rc[s[i+j]], rot*j)
 *   h       = canonical ? h_fw + h_rc (wrapping) : h_fw
 * Tables are data so that a seeded hasher is a parameter, not a rebuild; fw_xor / rc_xor (0 for
 * NtHasher) let the same form carry the alternative hashers of seq-hash (MulHasher, AntiLexHasher:
 * src/lib.rs:71-72, src/test.rs:81-83,107-109), whose arithmetic is NOT in the reference tree and has
 * no known-answer vector: PARITY UNPINNED - mmo_mul_hasher / mmo_antilex_hasher restate the published
 * idea with this repo's own constants. */
typedef struct mmo_hasher {
    uint32_t fw[4];
    uint32_t rc[4];
    uint32_t rot;
    uint32_t canonical;
    uint32_t fw_xor;
    uint32_t rc_xor;
    uint32_t kind; /* 0 NtHasher, 1 MulHasher, 2 AntiLexHasher (informational) */
} mmo_hasher;

enum { MMO_MINIMIZERS = 0, MMO_CLOSED_SYNCMERS = 1, MMO_OPEN_SYNCMERS = 2 };
enum { MMO_NAIVE = 0, MMO_STREAMING = 1 };

/* error codes (mirror the reference's assert!/panic! conditions) */
enum {
    MMO_OK = 0,
    MMO_ERR_W_ZERO = -1,          /* src/sliding_min.rs:91 */
    MMO_ERR_W_TOO_LARGE = -2,     /* src/sliding_min.rs:92-95 */
    MMO_ERR_LEN_TOO_LARGE = -3,   /* src/sliding_min.rs:96-99 */
    MMO_ERR_EVEN_L = -4,          /* src/canonical.rs:13-16 */
    MMO_ERR_HASHER_NOT_CANONICAL = -5, /* src/minimizers.rs:81,139 */
    MMO_ERR_OPEN_EVEN_W = -6,     /* src/syncmers.rs:24-29 */
    MMO_ERR_K_ZERO = -7,
    MMO_ERR_CAPACITY = -8,
    MMO_ERR_BAD_MODE = -9         /* src/lib.rs:437 */
};

/* Default NtHasher tables ("model M"). */
void mmo_default_hasher(mmo_hasher *h, int canonical);
/* MulHasher<CANONICAL>::new(k): "multiplies each character value by a pseudo-random constant"
 * (src/lib.rs:71-72).  PARITY UNPINNED: the constant and the character offset are this repo's. */
void mmo_mul_hasher(mmo_hasher *h, int canonical);
/* AntiLexHasher<CANONICAL>::new(k): the k-mer's own 2-bit value, first base most significant and
 * inverted (anti-lexicographic order), left-aligned in 32 bits.  PARITY UNPINNED. */
void mmo_antilex_hasher(mmo_hasher *h, uint32_t k, int canonical);

/* packed-seq layout: 4 bases/byte, base i at bits 2(i%4) of byte i/4; codes A0 C1 T2 G3 */
static inline uint32_t mmo_base(const uint8_t *packed, uint64_t i) {
    return (packed[i >> 2] >> (2 * (i & 3))) & 3u;
}
/* AsciiSeq / PackedSeqVec::from_ascii mapping: (c >> 1) & 3 */
void mmo_pack_ascii(const uint8_t *ascii, uint64_t n, uint8_t *packed /* ceil(n/4), zeroed by callee */);
/* Seq::to_revcomp: reversed order, code ^ 2 */
void mmo_revcomp_packed(const uint8_t *packed, uint64_t base_offset, uint64_t n, uint8_t *out);
/* deterministic synthetic generator G */
void mmo_gen_packed(uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *packed /* ceil(n/4) */);

/* all k-mer hashes of a sequence, closed form (one XOR chain per k-mer) */
int64_t mmo_hash_kmers_naive(const uint8_t *packed, uint64_t base_offset, uint64_t n, uint32_t k,
                             const mmo_hasher *h, uint32_t *out);
/* all k-mer hashes, rolling form */
int64_t mmo_hash_kmers_rolling(const uint8_t *packed, uint64_t base_offset, uint64_t n, uint32_t k,
                               const mmo_hasher *h, uint32_t *out);

/* per-window minimizer positions (one per window, before dedup).
 * flavour NAIVE: definition (src/minimizers.rs:22-28 + src/canonical.rs:18-29)
 * flavour STREAMING: two-stacks restatement (src/sliding_min.rs:86-212,
 *                    src/minimizers.rs:38-49,74-129) */
int64_t mmo_window_positions(const uint8_t *packed, uint64_t base_offset, uint64_t n, uint32_t k,
                             uint32_t w, const mmo_hasher *h, int canonical_windows, int flavour,
                             uint32_t *out /* n_w */);

/* collectors (src/collect.rs:15-76, src/syncmers.rs:19-48) */
uint64_t mmo_collect_and_dedup(const uint32_t *in, uint64_t n, uint32_t *out);
uint64_t mmo_collect_and_dedup_with_index(const uint32_t *in, uint64_t n, uint32_t *out,
                                          uint32_t *idx);
int64_t mmo_collect_syncmers(const uint32_t *in, uint64_t n, uint32_t w, int open, uint32_t *out);

/* The whole path = Builder::run_scalar (src/lib.rs:386-448, :504-537).
 * Returns the number of outputs (>= 0) or a negative MMO_ERR_*.
 * out_sk may be NULL; it is only legal with mode == MMO_MINIMIZERS. */
int64_t mmo_run(const uint8_t *packed, uint64_t base_offset, uint64_t n, uint32_t k, uint32_t w,
                const mmo_hasher *h, int canonical_windows, int mode, int flavour,
                uint32_t *out_pos, uint32_t *out_sk, uint64_t cap);

/* One-pass, optionally multi-threaded port of the minimizer path (mode 0), used as the timed
 * cpu_baseline of bench.py; same output as mmo_run(..., MMO_MINIMIZERS, ...). */
/* lanes per thread of mmo_run_fast in this build: 8 (AVX2, -march=native builds) or 1 (scalar) */
int mmo_fast_lanes(void);
int64_t mmo_run_fast(const uint8_t *packed, uint64_t base_offset, uint64_t n, uint32_t k, uint32_t w,
                     const mmo_hasher *h, int canonical_windows, int threads, uint32_t *out_pos,
                     uint64_t cap);

/* Output::values_u64 (src/lib.rs:579-612) + read_kmer / read_revcomp_kmer (packed-seq) */
uint64_t mmo_read_kmer_u64(const uint8_t *packed, uint64_t base_offset, uint32_t len, uint64_t pos);
uint64_t mmo_read_revcomp_kmer_u64(const uint8_t *packed, uint64_t base_offset, uint32_t len,
                                   uint64_t pos);
void mmo_values_u64(const uint8_t *packed, uint64_t base_offset, uint32_t len, int canonical,
                    const uint32_t *pos, uint64_t n_pos, uint64_t *out);

void mmo_values_u128(const uint8_t *packed, uint64_t base_offset, uint32_t len, int canonical,
                     const uint32_t *pos, uint64_t n_pos, uint64_t *out /* 2 per value */);

/* ---- skip-ambiguous windows (PackedNSeq): src/minimizers.rs:18-19,169-214, src/lib.rs:451-496.
 * A PackedNSeq is a PackedSeq plus one ambiguity bit per base (packed-seq 5.0.0 BitSeq, not in
 * tree; [INFERRED] layout: base i at bit i%8 of byte i/8, set for every non-ACGT character,
 * while the 2-bit code of such a character is the lossy (c>>1)&3).  A window with any ambiguous
 * base among its l = k+w-1 bases yields SKIPPED = u32::MAX-1 instead of a position. */
#define MMO_SKIPPED 0xFFFFFFFEu
void mmo_pack_ascii_n(const uint8_t *ascii, uint64_t n, uint8_t *packed /* ceil(n/4) */,
                      uint8_t *amb /* ceil(n/8) */);
/* per-window stream of canonical_minimizers_skip_ambiguous_windows (src/minimizers.rs:169-214):
 * the window's position, or MMO_SKIPPED where the l-mer holds an ambiguous base (:203-212) */
int64_t mmo_window_positions_skip_ambiguous(const uint8_t *packed, uint64_t base_offset,
                                            const uint8_t *amb, uint64_t amb_offset, uint64_t n,
                                            uint32_t k, uint32_t w, const mmo_hasher *h,
                                            int canonical_windows, uint32_t *out /* n_w */);
/* collect_and_dedup_into::<SKIP_MAX> on one lane's stream.  rule 0 = the AVX2/NEON kernels
 * (src/intrinsics/dedup.rs:133-159: compare with the IMMEDIATE predecessor, then drop SKIPPED),
 * rule 1 = the portable kernel and the scalar tail (src/intrinsics/dedup.rs:28-50,
 * src/collect.rs:243: compare with the LAST EMITTED value).  Pinned by src/test.rs:358-399. */
uint64_t mmo_collect_and_dedup_skip(const uint32_t *in, uint64_t n, int skip_max, int rule,
                                    uint32_t *out);
/* Builder::run_skip_ambiguous_windows (src/lib.rs:451-496): modes 0/1/2 as mmo_run. */
int64_t mmo_run_skip_ambiguous(const uint8_t *packed, uint64_t base_offset, const uint8_t *amb,
                               uint64_t amb_offset, uint64_t n, uint32_t k, uint32_t w,
                               const mmo_hasher *h, int canonical_windows, int mode, int rule,
                               uint32_t *out_pos, uint64_t cap);

/* order-sensitive checksum used by the large-size parity tests */
void mmo_checksum(const uint32_t *v, uint64_t n, uint64_t *weighted, uint64_t *plain);

#ifdef __cplusplus
}
#endif
#endif
