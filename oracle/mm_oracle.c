/*
 * mm_oracle.c — CPU restatement of the simd-minimizers hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see mm_oracle.h).  Never linked into the product.
 *
 * The reference is rust-seq/simd-minimizers v3.0.0; citations are
 * path:line relative to /root/reference.  The hash arithmetic (seq-hash
 * 0.2.0) and the PackedSeq layout (packed-seq 5.0.0) are third-party crates
 * that are NOT in the reference tree: their published behaviour is restated
 * here and anchored on the reference's doctest / unit-test vectors.
 */
#define _POSIX_C_SOURCE 200809L /* pthread barriers under -std=c11 */
#include "mm_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline uint32_t rotl32(uint32_t x, uint32_t r) {
    r &= 31u;
    return r ? (x << r) | (x >> (32u - r)) : x;
}
static inline uint32_t rotr32(uint32_t x, uint32_t r) { return rotl32(x, 32u - (r & 31u)); }

/* seq-hash 0.2.0 NtHasher::new(k): low 32 bits of the classic ntHash seeds
 * (constants: bench/src/nthash.rs:24-32), indexed by the packed 2-bit code
 * in the order the classic table lists them; 7 bits of rotation per base;
 * complement = code ^ 2; canonical combine = wrapping add. */
void mmo_default_hasher(mmo_hasher *h, int canonical) {
    static const uint32_t F[4] = {0x95c60474u, 0x62a02b4cu, 0x82572324u, 0x4be24456u};
    for (int c = 0; c < 4; ++c) {
        h->fw[c] = F[c];
        h->rc[c] = F[c ^ 2];
    }
    h->rot = 7;
    h->canonical = canonical ? 1u : 0u;
    h->fw_xor = h->rc_xor = 0;
    h->kind = 0;
}

/* PARITY UNPINNED (seq-hash MulHasher is not in the reference tree): character value (code + 1) times
 * the 32-bit golden-ratio constant, one bit of rotation per base... the structure is the rolling
 * rot-xor form of NtHasher with the table look-up replaced by a multiplication (src/lib.rs:71-72). */
void mmo_mul_hasher(mmo_hasher *h, int canonical) {
    for (uint32_t c = 0; c < 4; ++c) {
        h->fw[c] = (c + 1u) * 0x9E3779B1u;
        h->rc[c] = ((c ^ 2u) + 1u) * 0x9E3779B1u;
    }
    h->rot = 7;
    h->canonical = canonical ? 1u : 0u;
    h->fw_xor = h->rc_xor = 0;
    h->kind = 1;
}

/* PARITY UNPINNED (seq-hash AntiLexHasher is not in the reference tree): h_fw = the k-mer read as a
 * base-4 number, first base most significant, left-aligned in 32 bits (for k > 16 the older bases
 * wrap around and overlap), with the FIRST base inverted - the anti-lexicographic order; h_rc = the
 * same of the reverse complement. */
void mmo_antilex_hasher(mmo_hasher *h, uint32_t k, int canonical) {
    const uint32_t sh = (32u - ((2u * k) & 31u)) & 31u; /* base j of the k-mer lands at bits 2(k-1-j) + sh */
    for (uint32_t c = 0; c < 4; ++c) {
        h->fw[c] = rotl32(c, sh);
        /* reverse complement: base j of the window is base k-1-j of the rc k-mer, i.e. rotation 2j + sh */
        h->rc[c] = rotl32(c ^ 2u, sh);
    }
    h->rot = 2;
    h->canonical = canonical ? 1u : 0u;
    /* the first base of the forward k-mer sits at rotation 2(k-1) + sh = bits 30..31 (mod 32), and so
     * does the first base of the reverse-complement k-mer (window base k-1, rotation 2(k-1) + sh) */
    h->fw_xor = 3u << 30;
    h->rc_xor = 3u << 30;
    h->kind = 2;
}

void mmo_pack_ascii(const uint8_t *ascii, uint64_t n, uint8_t *packed) {
    memset(packed, 0, (size_t)((n + 3) / 4));
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t code = (ascii[i] >> 1) & 3u;
        packed[i >> 2] |= (uint8_t)(code << (2 * (i & 3)));
    }
}

void mmo_revcomp_packed(const uint8_t *packed, uint64_t base_offset, uint64_t n, uint8_t *out) {
    memset(out, 0, (size_t)((n + 3) / 4));
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t code = mmo_base(packed, base_offset + (n - 1 - i)) ^ 2u;
        out[i >> 2] |= (uint8_t)(code << (2 * (i & 3)));
    }
}

static inline uint64_t splitmix_final(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

void mmo_gen_packed(uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *packed) {
    const uint64_t g = 0x9E3779B97F4A7C15ull;
    memset(packed, 0, (size_t)((n + 3) / 4));
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t z = splitmix_final((first_base + i) + seed * g + g);
        uint32_t code = (uint32_t)(z >> 62);
        packed[i >> 2] |= (uint8_t)(code << (2 * (i & 3)));
    }
}

/* ------------------------------------------------------------------ hash */

int64_t mmo_hash_kmers_naive(const uint8_t *packed, uint64_t off, uint64_t n, uint32_t k,
                             const mmo_hasher *h, uint32_t *out) {
    if (k == 0) return MMO_ERR_K_ZERO;
    if (n < k) return 0;
    uint64_t nk = n - k + 1;
    for (uint64_t i = 0; i < nk; ++i) {
        uint32_t fw = 0, rc = 0;
        for (uint32_t j = 0; j < k; ++j) {
            uint32_t c = mmo_base(packed, off + i + j);
            fw ^= rotl32(h->fw[c], h->rot * (k - 1 - j));
            rc ^= rotl32(h->rc[c], h->rot * j);
        }
        fw ^= h->fw_xor;
        rc ^= h->rc_xor;
        out[i] = h->canonical ? fw + rc : fw;
    }
    return (int64_t)nk;
}

typedef struct {
    uint32_t fw, rc;
    uint32_t fw_out[4]; /* rotl(fw[c], rot*k)       : leaving base, forward strand */
    uint32_t rc_in[4];  /* rotl(rc[c], rot*(k-1))   : entering base, reverse strand */
    const mmo_hasher *h;
} roll_state;

static void roll_init(roll_state *st, const mmo_hasher *h, uint32_t k) {
    st->fw = st->rc = 0;
    st->h = h;
    for (int c = 0; c < 4; ++c) {
        st->fw_out[c] = rotl32(h->fw[c], h->rot * k);
        st->rc_in[c] = rotl32(h->rc[c], h->rot * (k - 1));
    }
}
/* add-only step used while the first k-1 bases are consumed
 * (the `take(delay1)` / `take(k-1-delay1)` warm-ups of src/minimizers.rs:97-108) */
static inline void roll_push(roll_state *st, uint32_t in) {
    st->fw = rotl32(st->fw, st->h->rot) ^ st->h->fw[in];
    st->rc = rotr32(st->rc, st->h->rot) ^ st->rc_in[in];
}
/* in/out step = hasher.in_out_mapper_scalar (call site src/minimizers.rs:85,110,121) */
static inline void roll_step(roll_state *st, uint32_t in, uint32_t out) {
    st->fw = rotl32(st->fw, st->h->rot) ^ st->fw_out[out] ^ st->h->fw[in];
    st->rc = rotr32(st->rc ^ st->h->rc[out], st->h->rot) ^ st->rc_in[in];
}
static inline uint32_t roll_value(const roll_state *st) {
    const uint32_t fw = st->fw ^ st->h->fw_xor, rc = st->rc ^ st->h->rc_xor;
    return st->h->canonical ? fw + rc : fw;
}

int64_t mmo_hash_kmers_rolling(const uint8_t *packed, uint64_t off, uint64_t n, uint32_t k,
                               const mmo_hasher *h, uint32_t *out) {
    if (k == 0) return MMO_ERR_K_ZERO;
    if (n < k) return 0;
    roll_state st;
    roll_init(&st, h, k);
    for (uint32_t j = 0; j < k; ++j) roll_push(&st, mmo_base(packed, off + j));
    uint64_t nk = n - k + 1;
    out[0] = roll_value(&st);
    for (uint64_t i = 1; i < nk; ++i) {
        roll_step(&st, mmo_base(packed, off + i + k - 1), mmo_base(packed, off + i - 1));
        out[i] = roll_value(&st);
    }
    return (int64_t)nk;
}

/* --------------------------------------------------- per-window positions */

static int check_params(uint64_t n, uint32_t k, uint32_t w, const mmo_hasher *h, int canonical) {
    if (k == 0) return MMO_ERR_K_ZERO;
    if (w == 0) return MMO_ERR_W_ZERO;
    if (w >= (1u << 15)) return MMO_ERR_W_TOO_LARGE;
    if (n >= (1ull << 32)) return MMO_ERR_LEN_TOO_LARGE;
    if (canonical) {
        if (!h->canonical) return MMO_ERR_HASHER_NOT_CANONICAL;
        if (((uint64_t)k + w - 1) % 2 == 0) return MMO_ERR_EVEN_L;
    }
    return MMO_OK;
}

/* Definition: src/minimizers.rs:22-28 (leftmost argmin of hash & 0xffff0000) and, for canonical
 * windows, src/canonical.rs:18-29 (#TG > l/2 -> leftmost, else rightmost: src/minimizers.rs:125). */
static int64_t positions_naive(const uint8_t *packed, uint64_t off, uint64_t n, uint32_t k,
                               uint32_t w, const mmo_hasher *h, int canonical, uint32_t *out) {
    uint64_t l = (uint64_t)k + w - 1;
    if (n < l) return 0;
    uint64_t nw = n - l + 1, nk = n - k + 1;
    uint32_t *hash = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nk);
    if (!hash) return MMO_ERR_CAPACITY;
    mmo_hash_kmers_naive(packed, off, n, k, h, hash);
    for (uint64_t i = 0; i < nw; ++i) {
        uint32_t best = hash[i] & 0xffff0000u;
        uint64_t left = i, right = i;
        for (uint64_t j = i + 1; j < i + w; ++j) {
            uint32_t v = hash[j] & 0xffff0000u;
            if (v < best) {
                best = v;
                left = right = j;
            } else if (v == best) {
                right = j;
            }
        }
        uint64_t pos = left;
        if (canonical) {
            uint64_t tg = 0;
            for (uint64_t j = i; j < i + l; ++j) tg += (mmo_base(packed, off + j) >> 1) & 1u;
            pos = (2 * tg > l) ? left : right;
        }
        out[i] = (uint32_t)pos;
    }
    free(hash);
    return (int64_t)nw;
}

/* Two-stacks sliding minimum, src/sliding_min.rs:86-142 (LEFT) and :145-212 (left+right).
 * Elements are (hash & 0xffff0000) | pos16; the 16-bit position is rebased by
 * delta = 65534 - w whenever it reaches 65535 (:117-125, :184-194). */
typedef struct {
    uint32_t w, idx, pos, pos_offset;
    uint32_t pl, pr;   /* prefix minima (left: min, right: max of !hash) */
    uint32_t *rl, *rr; /* ring buffers of suffix minima */
} lrmin_state;

static int lrmin_init(lrmin_state *s, uint32_t w) {
    s->w = w;
    s->idx = 0;
    s->pos = 0;
    s->pos_offset = 0;
    s->pl = s->pr = 0xffffffffu;
    s->rl = (uint32_t *)malloc(sizeof(uint32_t) * w);
    s->rr = (uint32_t *)malloc(sizeof(uint32_t) * w);
    if (!s->rl || !s->rr) return -1;
    for (uint32_t i = 0; i < w; ++i) s->rl[i] = s->rr[i] = 0xffffffffu;
    return 0;
}
static void lrmin_free(lrmin_state *s) {
    free(s->rl);
    free(s->rr);
}
static inline void lrmin_push(lrmin_state *s, uint32_t val, uint32_t *left, uint32_t *right) {
    const uint32_t w = s->w;
    if (s->pos == 0xffffu) {
        uint32_t delta = (1u << 16) - 2u - w;
        s->pos -= delta;
        s->pl -= delta;
        s->pr -= delta;
        s->pos_offset += delta;
        for (uint32_t i = 0; i < w; ++i) {
            s->rl[i] -= delta;
            s->rr[i] -= delta;
        }
    }
    uint32_t le = (val & 0xffff0000u) | s->pos;
    uint32_t re = (~val & 0xffff0000u) | s->pos;
    s->pos += 1;
    s->rl[s->idx] = le;
    s->rr[s->idx] = re;
    if (++s->idx == w) s->idx = 0;
    if (le < s->pl) s->pl = le;
    if (re > s->pr) s->pr = re;
    if (s->idx == 0) {
        uint32_t sl = s->rl[w - 1], sr = s->rr[w - 1];
        for (uint32_t i = w - 1; i-- > 0;) {
            if (s->rl[i] < sl) sl = s->rl[i];
            if (s->rr[i] > sr) sr = s->rr[i];
            s->rl[i] = sl;
            s->rr[i] = sr;
        }
        s->pl = le;
        s->pr = re;
    }
    uint32_t ml = s->pl < s->rl[s->idx] ? s->pl : s->rl[s->idx];
    uint32_t mr = s->pr > s->rr[s->idx] ? s->pr : s->rr[s->idx];
    *left = (ml & 0xffffu) + s->pos_offset;
    *right = (mr & 0xffffu) + s->pos_offset;
}

/* src/minimizers.rs:38-49 (forward) and :74-129 (canonical), scalar flavour. */
static int64_t positions_streaming(const uint8_t *packed, uint64_t off, uint64_t n, uint32_t k,
                                   uint32_t w, const mmo_hasher *h, int canonical, uint32_t *out) {
    uint64_t l = (uint64_t)k + w - 1;
    if (n < l) return 0;
    uint64_t nw = n - l + 1;
    roll_state st;
    roll_init(&st, h, k);
    lrmin_state lr;
    if (lrmin_init(&lr, w)) return MMO_ERR_CAPACITY;
    /* strand counter of src/canonical.rs:12-31: cnt = -l; cnt += a&2; out = cnt>0; cnt -= r&2 */
    int64_t cnt = -(int64_t)l;
    uint64_t a = 0;
    /* first k-1 bases: hash warm-up only (src/minimizers.rs:97-108) */
    for (; a + 1 < k; ++a) {
        uint32_t c = mmo_base(packed, off + a);
        roll_push(&st, c);
        cnt += c & 2u;
    }
    uint32_t left, right;
    /* next w-1 bases: k-mers enter the sliding min, no complete window yet (:110-115) */
    for (; a + 1 < l; ++a) {
        uint32_t c = mmo_base(packed, off + a);
        if (a + 1 == k) roll_push(&st, c);
        else roll_step(&st, c, mmo_base(packed, off + a - k));
        cnt += c & 2u;
        lrmin_push(&lr, roll_value(&st), &left, &right);
    }
    /* one window per remaining base (:117-128) */
    for (uint64_t i = 0; i < nw; ++i, ++a) {
        uint32_t c = mmo_base(packed, off + a);
        if (a + 1 == k) roll_push(&st, c);
        else roll_step(&st, c, mmo_base(packed, off + a - k));
        cnt += c & 2u;
        int is_canonical = cnt > 0;
        cnt -= mmo_base(packed, off + a - (l - 1)) & 2u;
        lrmin_push(&lr, roll_value(&st), &left, &right);
        out[i] = canonical ? (is_canonical ? left : right) : left;
    }
    lrmin_free(&lr);
    return (int64_t)nw;
}

int64_t mmo_window_positions(const uint8_t *packed, uint64_t off, uint64_t n, uint32_t k,
                             uint32_t w, const mmo_hasher *h, int canonical, int flavour,
                             uint32_t *out) {
    int e = check_params(n, k, w, h, canonical);
    if (e) return e;
    return flavour == MMO_NAIVE ? positions_naive(packed, off, n, k, w, h, canonical, out)
                                : positions_streaming(packed, off, n, k, w, h, canonical, out);
}

/* ------------------------------------------------------------ collectors */

/* src/collect.rs:15-37: drop ADJACENT duplicates only. */
uint64_t mmo_collect_and_dedup(const uint32_t *in, uint64_t n, uint32_t *out) {
    if (n == 0) return 0;
    uint64_t m = 0;
    out[m++] = in[0];
    for (uint64_t i = 1; i < n; ++i)
        if (in[i] != in[i - 1]) out[m++] = in[i];
    return m;
}

/* src/collect.rs:39-76: idx[j] = index of the first window that selected out[j]. */
uint64_t mmo_collect_and_dedup_with_index(const uint32_t *in, uint64_t n, uint32_t *out,
                                          uint32_t *idx) {
    if (n == 0) return 0;
    uint64_t m = 0;
    out[0] = in[0];
    idx[0] = 0;
    m = 1;
    for (uint64_t i = 1; i < n; ++i)
        if (in[i] != in[i - 1]) {
            out[m] = in[i];
            idx[m] = (uint32_t)i;
            ++m;
        }
    return m;
}

/* src/syncmers.rs:19-48: emits WINDOW indices, no dedup. */
int64_t mmo_collect_syncmers(const uint32_t *in, uint64_t n, uint32_t w, int open, uint32_t *out) {
    if (open && (w % 2 == 0)) return MMO_ERR_OPEN_EVEN_W;
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t p = in[i];
        int keep = open ? (p == i + w / 2) : (p == i || p == i + w - 1);
        if (keep) out[m++] = (uint32_t)i;
    }
    return (int64_t)m;
}

/* ------------------------------------------------------------ whole path */

int64_t mmo_run(const uint8_t *packed, uint64_t off, uint64_t n, uint32_t k, uint32_t w,
                const mmo_hasher *h, int canonical, int mode, int flavour, uint32_t *out_pos,
                uint32_t *out_sk, uint64_t cap) {
    int e = check_params(n, k, w, h, canonical);
    if (e) return e;
    if (mode < 0 || mode > 2) return MMO_ERR_BAD_MODE;
    if (mode == MMO_OPEN_SYNCMERS && w % 2 == 0) return MMO_ERR_OPEN_EVEN_W;
    if (out_sk && mode != MMO_MINIMIZERS) return MMO_ERR_BAD_MODE; /* src/lib.rs:339,498-503 */
    uint64_t l = (uint64_t)k + w - 1;
    if (n < l) return 0;
    uint64_t nw = n - l + 1;
    uint32_t *win = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nw);
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nw);
    uint32_t *tmp2 = out_sk ? (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nw) : NULL;
    if (!win || !tmp || (out_sk && !tmp2)) {
        free(win);
        free(tmp);
        free(tmp2);
        return MMO_ERR_CAPACITY;
    }
    int64_t r = mmo_window_positions(packed, off, n, k, w, h, canonical, flavour, win);
    int64_t m = r;
    if (r >= 0) {
        if (mode == MMO_MINIMIZERS)
            m = out_sk ? (int64_t)mmo_collect_and_dedup_with_index(win, nw, tmp, tmp2)
                       : (int64_t)mmo_collect_and_dedup(win, nw, tmp);
        else
            m = mmo_collect_syncmers(win, nw, w, mode == MMO_OPEN_SYNCMERS, tmp);
        if (m >= 0) {
            if ((uint64_t)m > cap) m = MMO_ERR_CAPACITY;
            else {
                memcpy(out_pos, tmp, sizeof(uint32_t) * (size_t)m);
                if (out_sk) memcpy(out_sk, tmp2, sizeof(uint32_t) * (size_t)m);
            }
        }
    }
    free(win);
    free(tmp);
    free(tmp2);
    return m;
}

/* ------------------------------------------------- skip-ambiguous windows */

static inline int amb_bit(const uint8_t *amb, uint64_t i) { return (amb[i >> 3] >> (i & 7)) & 1; }

/* PackedNSeqVec::from_ascii (packed-seq, not in tree; [INFERRED]): lossy 2-bit code for every
 * character, ambiguity bit for everything that is not ACGT / acgt. */
void mmo_pack_ascii_n(const uint8_t *ascii, uint64_t n, uint8_t *packed, uint8_t *amb) {
    mmo_pack_ascii(ascii, n, packed);
    memset(amb, 0, (size_t)((n + 7) / 8));
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t c = ascii[i] & 0xDF; /* upper case */
        if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) amb[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
}

/* src/minimizers.rs:203-212: zip the position stream with the l-mer ambiguity stream
 * (par_iter_kmer_ambiguity(l, ..)) and blend SKIPPED in.  The min computation itself runs on
 * the lossy codes, exactly as for a plain PackedSeq. */
int64_t mmo_window_positions_skip_ambiguous(const uint8_t *packed, uint64_t off, const uint8_t *amb,
                                            uint64_t amb_off, uint64_t n, uint32_t k, uint32_t w,
                                            const mmo_hasher *h, int canonical, uint32_t *out) {
    int64_t r = mmo_window_positions(packed, off, n, k, w, h, canonical, MMO_STREAMING, out);
    if (r <= 0) return r;
    const uint64_t l = (uint64_t)k + w - 1, nw = (uint64_t)r;
    uint64_t cnt = 0; /* ambiguous bases in the current window */
    for (uint64_t j = 0; j < l; ++j) cnt += (uint64_t)amb_bit(amb, amb_off + j);
    for (uint64_t i = 0; i < nw; ++i) {
        if (cnt) out[i] = MMO_SKIPPED;
        if (i + 1 < nw) {
            cnt += (uint64_t)amb_bit(amb, amb_off + i + l);
            cnt -= (uint64_t)amb_bit(amb, amb_off + i);
        }
    }
    return r;
}

uint64_t mmo_collect_and_dedup_skip(const uint32_t *in, uint64_t n, int skip_max, int rule,
                                    uint32_t *out) {
    uint64_t m = 0;
    int have_prec = 0;
    uint32_t prec = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t cur = in[i];
        if (rule == 0) {
            /* dedup.rs:147-155: mask = (cur == predecessor) | (cur == SKIPPED) */
            const int dup = i > 0 && cur == in[i - 1];
            if (!dup && !(skip_max && cur == MMO_SKIPPED)) out[m++] = cur;
        } else {
            /* dedup.rs:41-48: prec only moves when something is emitted */
            if ((!have_prec || cur != prec) && !(skip_max && cur == MMO_SKIPPED)) {
                out[m++] = cur;
                prec = cur;
                have_prec = 1;
            }
        }
    }
    return m;
}

int64_t mmo_run_skip_ambiguous(const uint8_t *packed, uint64_t off, const uint8_t *amb,
                               uint64_t amb_off, uint64_t n, uint32_t k, uint32_t w,
                               const mmo_hasher *h, int canonical, int mode, int rule,
                               uint32_t *out_pos, uint64_t cap) {
    int e = check_params(n, k, w, h, canonical);
    if (e) return e;
    if (mode < 0 || mode > 2) return MMO_ERR_BAD_MODE;
    if (mode == MMO_OPEN_SYNCMERS && w % 2 == 0) return MMO_ERR_OPEN_EVEN_W;
    uint64_t l = (uint64_t)k + w - 1;
    if (n < l) return 0;
    uint64_t nw = n - l + 1;
    uint32_t *win = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nw);
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nw);
    if (!win || !tmp) {
        free(win);
        free(tmp);
        return MMO_ERR_CAPACITY;
    }
    int64_t m = mmo_window_positions_skip_ambiguous(packed, off, amb, amb_off, n, k, w, h, canonical, win);
    if (m >= 0) {
        if (mode == MMO_MINIMIZERS) {
            m = (int64_t)mmo_collect_and_dedup_skip(win, nw, 1, rule, tmp); /* src/lib.rs:476-477 */
        } else {
            /* src/syncmers.rs:113-120: SKIPPED never equals a window index; :154-164 tail keeps x < SKIPPED */
            const int open = mode == MMO_OPEN_SYNCMERS;
            uint64_t c = 0;
            for (uint64_t i = 0; i < nw; ++i) {
                const uint64_t p = win[i];
                if (p == MMO_SKIPPED) continue;
                if (open ? (p == i + w / 2) : (p == i || p == i + w - 1)) tmp[c++] = (uint32_t)i;
            }
            m = (int64_t)c;
        }
        if ((uint64_t)m > cap) m = MMO_ERR_CAPACITY;
        else memcpy(out_pos, tmp, sizeof(uint32_t) * (size_t)m);
    }
    free(win);
    free(tmp);
    return m;
}

/* ---------------------------------------------------------------- values */

/* packed-seq Seq::read_kmer: base j of the k-mer at bits 2j (little-endian), len <= 32. */
uint64_t mmo_read_kmer_u64(const uint8_t *packed, uint64_t off, uint32_t len, uint64_t pos) {
    uint64_t v = 0;
    for (uint32_t j = 0; j < len; ++j) v |= (uint64_t)mmo_base(packed, off + pos + j) << (2 * j);
    return v;
}
/* packed-seq Seq::read_revcomp_kmer: reversed order, code ^ 2. */
uint64_t mmo_read_revcomp_kmer_u64(const uint8_t *packed, uint64_t off, uint32_t len,
                                   uint64_t pos) {
    uint64_t v = 0;
    for (uint32_t j = 0; j < len; ++j)
        v |= (uint64_t)(mmo_base(packed, off + pos + (len - 1 - j)) ^ 2u) << (2 * j);
    return v;
}
/* src/lib.rs:598-611 */
void mmo_values_u64(const uint8_t *packed, uint64_t off, uint32_t len, int canonical,
                    const uint32_t *pos, uint64_t n_pos, uint64_t *out) {
    for (uint64_t i = 0; i < n_pos; ++i) {
        uint64_t a = mmo_read_kmer_u64(packed, off, len, pos[i]);
        if (canonical) {
            uint64_t b = mmo_read_revcomp_kmer_u64(packed, off, len, pos[i]);
            if (b < a) a = b;
        }
        out[i] = a;
    }
}

/* Output::values_u128 (src/lib.rs:613-629): len <= 64, out[2i] = low, out[2i+1] = high 64 bits */
void mmo_values_u128(const uint8_t *packed, uint64_t off, uint32_t len, int canonical,
                     const uint32_t *pos, uint64_t n_pos, uint64_t *out) {
    for (uint64_t i = 0; i < n_pos; ++i) {
        unsigned __int128 a = 0, b = 0;
        for (uint32_t j = 0; j < len; ++j) {
            a |= (unsigned __int128)mmo_base(packed, off + pos[i] + j) << (2 * j);
            b |= (unsigned __int128)(mmo_base(packed, off + pos[i] + (len - 1 - j)) ^ 2u) << (2 * j);
        }
        if (canonical && b < a) a = b;
        out[2 * i] = (uint64_t)a;
        out[2 * i + 1] = (uint64_t)(a >> 64);
    }
}

void mmo_checksum(const uint32_t *v, uint64_t n, uint64_t *weighted, uint64_t *plain) {
    uint64_t a = 0, b = 0;
    for (uint64_t j = 0; j < n; ++j) {
        a += (j + 1) * (uint64_t)v[j];
        b += v[j];
    }
    *weighted = a;
    *plain = b;
}

/* ------------------------------------------------ faster port for timing
 * Same algorithm as positions_streaming + collect_and_dedup (src/minimizers.rs:74-129,
 * src/sliding_min.rs:145-212, src/collect.rs:15-37), fused into one pass without the
 * per-window intermediate array, and optionally spread over threads by window range (like the
 * reference's rayon-over-contigs benchmark, bench/src/bin/paper.rs:442-459).  Used only as the
 * cpu_baseline of bench.py and checked against mmo_run by the tests. */
#include <pthread.h>

/* One output segment of a job: the job's output is the concatenation of its segments. */
typedef struct {
    uint32_t *p;
    uint64_t count, cap;
} fast_seg;

#define FAST_MAX_SEGS 10

typedef struct {
    const uint8_t *packed;
    uint64_t off, n;
    uint32_t k, w;
    const mmo_hasher *h;
    int canonical;
    uint64_t win_begin, win_end;
    uint32_t *out;          /* the job's slot: the segments live inside it */
    uint64_t cap, count;    /* slot size; total entries of all segments */
    fast_seg segs[FAST_MAX_SEGS];
    int n_segs, failed;
} fast_job;

/* scalar walk of windows [wb, we) into seg (dedup against window wb - 1 when it exists) */
static void scalar_range(const fast_job *jb, uint64_t wb, uint64_t we, fast_seg *seg) {
    const uint8_t *packed = jb->packed;
    const uint64_t off = jb->off;
    const uint32_t k = jb->k, w = jb->w;
    const uint64_t l = (uint64_t)k + w - 1;
    const int canonical = jb->canonical;
    seg->count = 0;
    if (wb >= we) return;
    /* element 0 = k-mer (wb - 1) when it exists, so that the seam dedup is exact */
    const uint64_t first_km = wb > 0 ? wb - 1 : 0;
    const int have_prev = wb > 0;
    roll_state st;
    roll_init(&st, jb->h, k);
    lrmin_state lr;
    if (lrmin_init(&lr, w)) {
        seg->count = seg->cap + 1; /* reported as a capacity failure */
        return;
    }
    uint64_t a = first_km; /* next base to consume */
    for (uint32_t j = 0; j + 1 < k; ++j, ++a) roll_push(&st, mmo_base(packed, off + a));
    int64_t cnt = 0;
    if (canonical) {
        /* T/G count of the first l-1 bases of the first window */
        cnt = -(int64_t)l;
        for (uint64_t j = first_km; j + 1 < first_km + l; ++j) cnt += mmo_base(packed, off + j) & 2u;
    }
    uint32_t left, right, prev = 0;
    int first = 1;
    uint64_t m = 0;
    /* k-mers first_km .. we + w - 2; window i completes with k-mer i + w - 1 */
    const uint64_t last_km = we + w - 1; /* exclusive */
    for (uint64_t km = first_km; km < last_km; ++km, ++a) {
        uint32_t c = mmo_base(packed, off + a);
        if (km == first_km) roll_push(&st, c);
        else roll_step(&st, c, mmo_base(packed, off + a - k));
        lrmin_push(&lr, roll_value(&st), &left, &right);
        if (km + 1 < first_km + w) continue; /* window not complete yet */
        uint64_t i = km + 1 - w;             /* window index */
        uint32_t pos = left;
        if (canonical) {
            cnt += c & 2u;
            pos = cnt > 0 ? left : right;
            cnt -= mmo_base(packed, off + i) & 2u;
        }
        pos += (uint32_t)first_km;
        if (i >= wb) {
            if ((first && !have_prev) || pos != prev) {
                if (m < seg->cap) seg->p[m] = pos;
                ++m;
            }
        }
        prev = pos;
        first = 0;
    }
    lrmin_free(&lr);
    seg->count = m;
}

#if defined(__AVX2__)
#include <immintrin.h>
/* Eight-lane flavour of the same walk for host-tuned builds (the reference's SIMD path also runs
 * eight chunks of the sequence side by side in the lanes of a 256-bit vector,
 * src/minimizers.rs:133-166, src/sliding_min.rs:302-355, src/collect.rs:128-285).  Written from
 * the scalar code above, not from the reference: lane L walks windows [vb + L * per8, + per8) with
 * its own predecessor window, so every lane is the scalar walk of a sub-range and the concatenation
 * of the eight outputs is exact.  per8 is a multiple of 16, which makes all lanes' base positions
 * congruent mod 16: one scalar shift count and one refill schedule serve all lanes of a stream. */
typedef struct {
    __m256i byteoff; /* byte offset of the current dword, per lane */
    __m256i cur;     /* current dword (16 bases), per lane */
    int sh;          /* bit position of the next base inside cur (same for all lanes) */
} vstream;

static inline void vstream_init(vstream *s, const uint8_t *packed, const uint64_t *pos /* [8] absolute bases */) {
    int32_t bo[8];
    for (int i = 0; i < 8; ++i) bo[i] = (int32_t)((pos[i] >> 4) << 2);
    s->byteoff = _mm256_loadu_si256((const __m256i *)bo);
    s->cur = _mm256_i32gather_epi32((const int *)packed, s->byteoff, 1);
    s->sh = (int)((pos[0] & 15u) * 2u);
}
static inline __m256i vstream_next(vstream *s, const uint8_t *packed) {
    const __m256i code = _mm256_and_si256(_mm256_srl_epi32(s->cur, _mm_cvtsi32_si128(s->sh)), _mm256_set1_epi32(3));
    s->sh += 2;
    if (s->sh == 32) {
        s->byteoff = _mm256_add_epi32(s->byteoff, _mm256_set1_epi32(4));
        s->cur = _mm256_i32gather_epi32((const int *)packed, s->byteoff, 1);
        s->sh = 0;
    }
    return code;
}
static inline __m256i vrotl(__m256i x, uint32_t r) {
    r &= 31u;
    if (r == 0) return x;
    return _mm256_or_si256(_mm256_sll_epi32(x, _mm_cvtsi32_si128((int)r)),
                           _mm256_srl_epi32(x, _mm_cvtsi32_si128((int)(32u - r))));
}
static inline __m256i vtable(const uint32_t t[4]) {
    return _mm256_setr_epi32((int)t[0], (int)t[1], (int)t[2], (int)t[3], (int)t[0], (int)t[1], (int)t[2], (int)t[3]);
}

/* windows [vb, vb + 8 * per8) of the job, vb > 0, per8 % 16 == 0; segs[0..7] receive the lanes */
static int avx2_range(const fast_job *jb, uint64_t vb, uint64_t per8, fast_seg *segs) {
    const uint8_t *packed = jb->packed;
    const uint32_t k = jb->k, w = jb->w;
    const uint64_t l = (uint64_t)k + w - 1;
    const mmo_hasher *h = jb->h;
    const int canonical = jb->canonical, hash_rc = h->canonical != 0;
    const uint32_t rot = h->rot;
    uint32_t fw_out[4], rc_in[4];
    for (int c = 0; c < 4; ++c) {
        fw_out[c] = rotl32(h->fw[c], rot * k);
        rc_in[c] = rotl32(h->rc[c], rot * (k - 1));
    }
    const __m256i T_fw = vtable(h->fw), T_fwout = vtable(fw_out), T_rc = vtable(h->rc), T_rcin = vtable(rc_in);
    __m256i *rl = (__m256i *)aligned_alloc(32, sizeof(__m256i) * w);
    __m256i *rr = (__m256i *)aligned_alloc(32, sizeof(__m256i) * w);
    if (!rl || !rr) {
        free(rl);
        free(rr);
        return -1;
    }
    const __m256i ones = _mm256_set1_epi32(-1), himask = _mm256_set1_epi32((int)0xffff0000u);
    for (uint32_t i = 0; i < w; ++i) rl[i] = rr[i] = ones;
    uint64_t first_km[8], posA[8];
    int32_t base_pos[8], cnt0[8];
    for (int L = 0; L < 8; ++L) {
        first_km[L] = vb + (uint64_t)L * per8 - 1;
        posA[L] = jb->off + first_km[L];
        base_pos[L] = (int32_t)first_km[L];
        int64_t c = -(int64_t)l;
        if (canonical)
            for (uint64_t j = first_km[L]; j + 1 < first_km[L] + l; ++j) c += mmo_base(packed, jb->off + j) & 2u;
        cnt0[L] = (int32_t)c;
    }
    vstream sa, sb, sc; /* entering base, base leaving the hash, base leaving the window */
    vstream_init(&sa, packed, posA);
    vstream_init(&sb, packed, posA);
    vstream_init(&sc, packed, posA);
    __m256i fw = _mm256_setzero_si256(), rc = _mm256_setzero_si256();
    for (uint32_t j = 0; j + 1 < k; ++j) { /* the first k - 1 bases: add only */
        const __m256i in = vstream_next(&sa, packed);
        fw = _mm256_xor_si256(vrotl(fw, rot), _mm256_permutevar8x32_epi32(T_fw, in));
        rc = _mm256_xor_si256(vrotl(rc, 32u - (rot & 31u)), _mm256_permutevar8x32_epi32(T_rcin, in));
    }
    __m256i cnt = _mm256_loadu_si256((const __m256i *)cnt0);
    const __m256i vbase = _mm256_loadu_si256((const __m256i *)base_pos);
    __m256i pl = ones, pr = ones, prev = _mm256_setzero_si256();
    uint32_t idx = 0, pos = 0, pos_offset = 0;
    uint64_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t steps = per8 + w; /* k-mers first_km .. first_km + per8 + w - 1 */
    for (uint64_t t = 0; t < steps; ++t) {
        const __m256i in = vstream_next(&sa, packed);
        if (t == 0) {
            fw = _mm256_xor_si256(vrotl(fw, rot), _mm256_permutevar8x32_epi32(T_fw, in));
            rc = _mm256_xor_si256(vrotl(rc, 32u - (rot & 31u)), _mm256_permutevar8x32_epi32(T_rcin, in));
        } else {
            const __m256i out = vstream_next(&sb, packed);
            fw = _mm256_xor_si256(_mm256_xor_si256(vrotl(fw, rot), _mm256_permutevar8x32_epi32(T_fwout, out)),
                                  _mm256_permutevar8x32_epi32(T_fw, in));
            rc = _mm256_xor_si256(vrotl(_mm256_xor_si256(rc, _mm256_permutevar8x32_epi32(T_rc, out)), 32u - (rot & 31u)),
                                  _mm256_permutevar8x32_epi32(T_rcin, in));
        }
        const __m256i fwx = _mm256_xor_si256(fw, _mm256_set1_epi32((int)h->fw_xor));
        const __m256i val = hash_rc ? _mm256_add_epi32(fwx, _mm256_xor_si256(rc, _mm256_set1_epi32((int)h->rc_xor))) : fwx;
        /* two-stacks minimum, same steps as lrmin_push */
        if (pos == 0xffffu) {
            const uint32_t delta = (1u << 16) - 2u - w;
            const __m256i d = _mm256_set1_epi32((int)delta);
            pos -= delta;
            pl = _mm256_sub_epi32(pl, d);
            pr = _mm256_sub_epi32(pr, d);
            pos_offset += delta;
            for (uint32_t i = 0; i < w; ++i) {
                rl[i] = _mm256_sub_epi32(rl[i], d);
                rr[i] = _mm256_sub_epi32(rr[i], d);
            }
        }
        const __m256i vpos = _mm256_set1_epi32((int)pos);
        const __m256i le = _mm256_or_si256(_mm256_and_si256(val, himask), vpos);
        const __m256i re = _mm256_or_si256(_mm256_andnot_si256(val, himask), vpos);
        pos += 1;
        rl[idx] = le;
        if (canonical) rr[idx] = re;
        if (++idx == w) idx = 0;
        pl = _mm256_min_epu32(pl, le);
        pr = _mm256_max_epu32(pr, re);
        if (idx == 0) {
            __m256i sl = rl[w - 1];
            for (uint32_t i = w - 1; i-- > 0;) {
                sl = _mm256_min_epu32(sl, rl[i]);
                rl[i] = sl;
            }
            if (canonical) { /* the rightmost minimum is only needed for the strand choice */
                __m256i sr = rr[w - 1];
                for (uint32_t i = w - 1; i-- > 0;) {
                    sr = _mm256_max_epu32(sr, rr[i]);
                    rr[i] = sr;
                }
            }
            pl = le;
            pr = re;
        }
        if (t + 1 < w) continue; /* window not complete yet */
        const __m256i ml = _mm256_min_epu32(pl, rl[idx]), mr = _mm256_max_epu32(pr, rr[idx]);
        const __m256i lo16 = _mm256_set1_epi32(0xffff), voff = _mm256_set1_epi32((int)pos_offset);
        __m256i p = _mm256_add_epi32(_mm256_and_si256(ml, lo16), voff);
        if (canonical) {
            const __m256i two = _mm256_set1_epi32(2);
            cnt = _mm256_add_epi32(cnt, _mm256_and_si256(in, two));
            const __m256i right = _mm256_add_epi32(_mm256_and_si256(mr, lo16), voff);
            p = _mm256_blendv_epi8(right, p, _mm256_cmpgt_epi32(cnt, _mm256_setzero_si256()));
            cnt = _mm256_sub_epi32(cnt, _mm256_and_si256(vstream_next(&sc, packed), two));
        }
        p = _mm256_add_epi32(p, vbase);
        if (t + 1 > w) { /* t + 1 == w is the predecessor window of every lane */
            unsigned flags = (unsigned)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(p, prev))) ^ 0xffu;
            if (flags) {
                uint32_t pv[8];
                _mm256_storeu_si256((__m256i *)pv, p);
                while (flags) {
                    const int L = __builtin_ctz(flags);
                    flags &= flags - 1;
                    if (m[L] < segs[L].cap) segs[L].p[m[L]] = pv[L];
                    ++m[L];
                }
            }
        }
        prev = p;
    }
    for (int L = 0; L < 8; ++L) segs[L].count = m[L];
    free(rl);
    free(rr);
    return 0;
}
#endif

/* A job's range as up to ten pieces: [scalar head] + eight vector lanes + [scalar tail] in
 * host-tuned builds, one scalar piece otherwise (MMO_NO_AVX2 forces the scalar walk). */
static int fast_range(fast_job *jb, fast_seg *segs, int *n_segs) {
    uint64_t wb = jb->win_begin, we = jb->win_end;
    uint32_t *p = jb->out;
    uint64_t room = jb->cap;
    *n_segs = 0;
#if defined(__AVX2__)
    if (!getenv("MMO_NO_AVX2") && we - wb >= 4096 && jb->n < (1ull << 31)) {
        const uint64_t vb = wb == 0 ? 16 : wb;          /* the vector lanes need a predecessor window */
        const uint64_t per8 = ((we - 64 - vb) / 8) & ~15ull; /* the last 64 windows stay scalar: no over-read */
        const uint64_t ve = vb + 8 * per8;
        if (per8 >= 16) {
            const double dens = (double)jb->cap / (double)(we - wb + 1);
            if (vb > wb) { /* scalar head [0, 16) */
                fast_seg *sg = &segs[(*n_segs)++];
                sg->p = p; sg->cap = room < 32 ? room : 32;
                scalar_range(jb, wb, vb, sg);
                p += sg->cap; room -= sg->cap;
            }
            uint64_t lane_cap = (uint64_t)(dens * (double)per8) + 64;
            /* the scalar tail holds fewer than 64 + 8 * 16 windows */
            if (lane_cap * 8 + 256 > room) lane_cap = room > 256 ? (room - 256) / 8 : 0;
            fast_seg *lanes = &segs[*n_segs];
            for (int L = 0; L < 8; ++L) {
                lanes[L].p = p + (uint64_t)L * lane_cap;
                lanes[L].cap = lane_cap;
                lanes[L].count = 0;
            }
            if (avx2_range(jb, vb, per8, lanes)) return -1;
            *n_segs += 8;
            p += 8 * lane_cap; room -= 8 * lane_cap;
            wb = ve;
        }
    }
#endif
    fast_seg *sg = &segs[(*n_segs)++];
    sg->p = p; sg->cap = room;
    scalar_range(jb, wb, we, sg);
    return 0;
}

static void fast_job_run(fast_job *jb) {
    jb->failed = fast_range(jb, jb->segs, &jb->n_segs) != 0;
    jb->count = 0;
    for (int g = 0; g < jb->n_segs; ++g) {
        if (jb->segs[g].count > jb->segs[g].cap) jb->failed = 1;
        jb->count += jb->segs[g].count;
    }
}

/* Worker of mmo_run_fast: walk the range into the thread's slot, wait for everybody, let worker 0
 * turn the counts into offsets, then copy the slot to its place in the caller's array (the merge runs
 * in parallel too: a serial merge of 180 MB per 256 Mbp would dominate with many threads). */
typedef struct fast_shared {
    pthread_barrier_t bar;
    fast_job *jobs;
    int threads;
    uint32_t *out_pos;
    uint64_t cap;
    uint64_t *dst;   /* [threads] first output slot of every worker */
    int64_t total;
} fast_shared;

typedef struct fast_arg {
    fast_shared *sh;
    int t;
} fast_arg;

static void *fast_thread(void *p) {
    fast_arg *fa = (fast_arg *)p;
    fast_shared *sh = fa->sh;
    fast_job *jb = &sh->jobs[fa->t];
    fast_job_run(jb);
    pthread_barrier_wait(&sh->bar);
    if (fa->t == 0) {
        uint64_t m = 0;
        int fits = 1;
        for (int t = 0; t < sh->threads; ++t) {
            if (sh->jobs[t].failed) fits = 0;
            sh->dst[t] = m;
            m += sh->jobs[t].count;
        }
        sh->total = (fits && m <= sh->cap) ? (int64_t)m : MMO_ERR_CAPACITY;
    }
    pthread_barrier_wait(&sh->bar);
    if (sh->total >= 0) {
        uint32_t *dst = sh->out_pos + sh->dst[fa->t];
        for (int g = 0; g < jb->n_segs; ++g) {
            memcpy(dst, jb->segs[g].p, sizeof(uint32_t) * (size_t)jb->segs[g].count);
            dst += jb->segs[g].count;
        }
    }
    return NULL;
}

/* Per-worker output slots are kept between calls (grow-only), like the reference's reusable
 * thread-local buffers (src/lib.rs:80-81,217-219): a fresh 8 MB malloc per worker and call would put
 * tens of thousands of page faults into every timed call. */
static uint32_t *g_fast_slots = NULL;
static uint64_t g_fast_slots_elems = 0;

/* 1 when mmo_run_fast of this build walks eight lanes per thread with AVX2, 0 for the scalar walk */
int mmo_fast_lanes(void) {
#if defined(__AVX2__)
    return getenv("MMO_NO_AVX2") ? 1 : 8;
#else
    return 1;
#endif
}

/* Minimizer positions (mode 0) with `threads` worker threads; returns the count or MMO_ERR_*. */
int64_t mmo_run_fast(const uint8_t *packed, uint64_t off, uint64_t n, uint32_t k, uint32_t w,
                     const mmo_hasher *h, int canonical, int threads, uint32_t *out_pos, uint64_t cap) {
    int e = check_params(n, k, w, h, canonical);
    if (e) return e;
    uint64_t l = (uint64_t)k + w - 1;
    if (n < l) return 0;
    uint64_t nw = n - l + 1;
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > nw) threads = (int)nw;
    fast_job *jobs = (fast_job *)calloc((size_t)threads, sizeof(fast_job));
    uint64_t per = (nw + threads - 1) / threads;
    uint64_t slot = (uint64_t)((double)per * 2.2 / (w + 1.0)) + 4096;
    if (slot > per + 4096) slot = per + 4096; /* a window emits at most once; room for the piece layout */
    if (slot < 4096) slot = 4096;
    if (g_fast_slots_elems < slot * (uint64_t)threads) {
        free(g_fast_slots);
        g_fast_slots_elems = slot * (uint64_t)threads;
        g_fast_slots = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)g_fast_slots_elems);
        if (!g_fast_slots) {
            g_fast_slots_elems = 0;
            free(jobs);
            return MMO_ERR_CAPACITY;
        }
    }
    for (int t = 0; t < threads; ++t) {
        fast_job *jb = &jobs[t];
        jb->packed = packed; jb->off = off; jb->n = n; jb->k = k; jb->w = w; jb->h = h;
        jb->canonical = canonical;
        jb->win_begin = (uint64_t)t * per < nw ? (uint64_t)t * per : nw;
        jb->win_end = (uint64_t)(t + 1) * per < nw ? (uint64_t)(t + 1) * per : nw;
        jb->cap = slot;
        jb->out = g_fast_slots + (uint64_t)t * slot;
    }
    int64_t total;
    if (threads == 1) {
        /* one worker: walk into a scratch slot as well (the pieces are then packed into out_pos) */
        fast_job_run(&jobs[0]);
        total = MMO_ERR_CAPACITY;
        if (!jobs[0].failed && jobs[0].count <= cap) {
            uint32_t *dst = out_pos;
            for (int g = 0; g < jobs[0].n_segs; ++g) {
                memcpy(dst, jobs[0].segs[g].p, sizeof(uint32_t) * (size_t)jobs[0].segs[g].count);
                dst += jobs[0].segs[g].count;
            }
            total = (int64_t)jobs[0].count;
        }
    } else {
        fast_shared sh;
        pthread_barrier_init(&sh.bar, NULL, (unsigned)threads);
        sh.jobs = jobs;
        sh.threads = threads;
        sh.out_pos = out_pos;
        sh.cap = cap;
        sh.dst = (uint64_t *)calloc((size_t)threads, sizeof(uint64_t));
        sh.total = MMO_ERR_CAPACITY;
        pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
        fast_arg *args = (fast_arg *)calloc((size_t)threads, sizeof(fast_arg));
        for (int t = 0; t < threads; ++t) {
            args[t].sh = &sh;
            args[t].t = t;
            pthread_create(&tids[t], NULL, fast_thread, &args[t]);
        }
        for (int t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
        total = sh.total;
        pthread_barrier_destroy(&sh.bar);
        free(sh.dst);
        free(tids);
        free(args);
    }
    free(jobs);
    return total;
}
