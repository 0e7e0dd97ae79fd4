// simd_minimizers_amd.hpp — header-only C++ mirror of the reference's builder API
// (rust-seq/simd-minimizers src/lib.rs:225-654) over the C ABI of simd_minimizers_amd.h.
//
//   using namespace simd_minimizers;
//   std::vector<uint32_t> pos, sk;
//   auto out = canonical_minimizers(21, 11).super_kmers(&sk).run(PackedSeq{bytes, 0, n}, pos);
//   std::vector<uint64_t> vals = out.values_u64();
//
// Names, argument meaning and error behaviour follow the reference: `run` APPENDS to the
// output vector (src/lib.rs:80-81) and drops a leading result equal to its previous last
// element (src/collect.rs:265-271); the reference's assert!/panic! conditions surface as
// simd_minimizers::Error carrying the MM_ERR_* code.  All compute happens in the HIP library.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "simd_minimizers_amd.h"

namespace simd_minimizers {

struct Error : std::runtime_error {
    int code;
    Error(int c) : std::runtime_error(std::string(mm_strerror(c)) + " " + mm_last_error()), code(c) {}
};
inline void check(int c) {
    if (c != MM_OK) throw Error(c);
}

// packed-seq PackedSeq: borrowed view of 2-bit bases (A0 C1 T2 G3, 4 per byte)
struct PackedSeq {
    const uint8_t *data;
    uint64_t offset;  // first base inside `data`
    uint64_t len;
    PackedSeq slice(uint64_t b, uint64_t e) const { return PackedSeq{data, offset + b, e - b}; }
};
// packed-seq PackedNSeq: a PackedSeq plus one ambiguity bit per base (bit (amb_offset+i)%8 of byte
// (amb_offset+i)/8 of `amb`)
struct PackedNSeq {
    PackedSeq seq;
    const uint8_t *amb;
    uint64_t amb_offset;
    PackedNSeq slice(uint64_t b, uint64_t e) const { return PackedNSeq{seq.slice(b, e), amb, amb_offset + b}; }
};
// packed-seq AsciiSeq: ACTG / actg characters
struct AsciiSeq {
    const uint8_t *data;
    uint64_t len;
};

// seq-hash NtHasher<CANONICAL>::new(k)
template <bool CANONICAL = true>
struct NtHasher {
    mm_hasher_t tables;
    uint32_t k;
    explicit NtHasher(uint32_t k_) : k(k_) { check(mm_default_hasher(&tables, CANONICAL)); }
    bool is_canonical() const { return CANONICAL; }
};
// seq-hash MulHasher<CANONICAL>::new(k) / AntiLexHasher<CANONICAL>::new(k) (src/lib.rs:71-72; src/test.rs:81-83,
// 107-109).  PARITY UNPINNED: their arithmetic is not in the reference tree (see mm_mul_hasher /
// mm_antilex_hasher in the C header); a caller who knows the real per-base values fills `tables` itself.
template <bool CANONICAL = true>
struct MulHasher {
    mm_hasher_t tables;
    uint32_t k;
    explicit MulHasher(uint32_t k_) : k(k_) { check(mm_mul_hasher(&tables, CANONICAL)); }
    bool is_canonical() const { return CANONICAL; }
};
template <bool CANONICAL = true>
struct AntiLexHasher {
    mm_hasher_t tables;
    uint32_t k;
    explicit AntiLexHasher(uint32_t k_) : k(k_) { check(mm_antilex_hasher(&tables, k_, CANONICAL)); }
    bool is_canonical() const { return CANONICAL; }
};

class Workspace {
  public:
    explicit Workspace(int device = 0, void *stream = nullptr) { check(mm_workspace_create(&ws_, device, stream)); }
    ~Workspace() { mm_workspace_destroy(ws_); }
    Workspace(const Workspace &) = delete;
    Workspace &operator=(const Workspace &) = delete;
    mm_workspace_t *get() const { return ws_; }
    static Workspace &thread_default() {  // the reference's thread_local CACHE (src/lib.rs:217-219)
        thread_local Workspace w;
        return w;
    }

  private:
    mm_workspace_t *ws_ = nullptr;
};

using u128 = unsigned __int128;

template <bool CANONICAL>
class Output {  // src/lib.rs:232-237
  public:
    Output(uint32_t len, PackedSeq seq, const std::vector<uint32_t> *pos, Workspace *ws)
        : len_(len), seq_(seq), pos_(pos), ws_(ws) {}
    std::vector<uint64_t> values_u64() const {  // src/lib.rs:584-612
        std::vector<uint64_t> v(pos_->size());
        check(mm_values_u64_host(ws_->get(), seq_.data, seq_.offset, seq_.len, len_, CANONICAL, pos_->data(),
                                 pos_->size(), v.data()));
        return v;
    }
    std::vector<u128> values_u128() const {  // src/lib.rs:587-593 (len <= 64)
        std::vector<uint64_t> raw(2 * pos_->size());
        check(mm_values_u128_host(ws_->get(), seq_.data, seq_.offset, seq_.len, len_, CANONICAL, pos_->data(),
                                  pos_->size(), raw.data()));
        std::vector<u128> v(pos_->size());
        for (size_t i = 0; i < v.size(); ++i) v[i] = ((u128)raw[2 * i + 1] << 64) | raw[2 * i];
        return v;
    }
    std::vector<std::pair<uint32_t, uint64_t>> pos_and_values_u64() const {  // src/lib.rs:598-612
        const std::vector<uint64_t> v = values_u64();
        std::vector<std::pair<uint32_t, uint64_t>> r(v.size());
        for (size_t i = 0; i < v.size(); ++i) r[i] = {(*pos_)[i], v[i]};
        return r;
    }
    std::vector<std::pair<uint32_t, u128>> pos_and_values_u128() const {  // src/lib.rs:615-630
        const std::vector<u128> v = values_u128();
        std::vector<std::pair<uint32_t, u128>> r(v.size());
        for (size_t i = 0; i < v.size(); ++i) r[i] = {(*pos_)[i], v[i]};
        return r;
    }
    const std::vector<uint32_t> &positions() const { return *pos_; }

  private:
    uint32_t len_;
    PackedSeq seq_;
    const std::vector<uint32_t> *pos_;
    Workspace *ws_;
};

template <bool CANONICAL, int SYNCMER>
class Builder {  // src/lib.rs:225-230
  public:
    Builder(uint32_t k, uint32_t w) : k_(k), w_(w) {}
    template <class H>
    Builder hasher(const H &h) const {  // src/lib.rs:327 (any KmerHasher: NtHasher, MulHasher, AntiLexHasher)
        Builder b = *this;
        b.hasher_ = h.tables;
        b.has_hasher_ = true;
        return b;
    }
    Builder super_kmers(std::vector<uint32_t> *sk) const {  // src/lib.rs:341 (minimizers only)
        static_assert(SYNCMER == 0, "super_kmers() is only defined for minimizers");
        Builder b = *this;
        b.sk_ = sk;
        return b;
    }
    Builder workspace(Workspace *ws) const {
        Builder b = *this;
        b.ws_ = ws;
        return b;
    }
    Output<CANONICAL> run(PackedSeq seq, std::vector<uint32_t> &min_pos) const {  // src/lib.rs:378
        Workspace &ws = ws_ ? *ws_ : Workspace::thread_default();
        std::vector<uint32_t> pos, sk;
        const uint64_t n = compute(seq, ws, pos, sk);
        size_t first = 0;
        if (SYNCMER == 0)
            while (first < n && !min_pos.empty() && pos[first] == min_pos.back()) ++first;
        min_pos.insert(min_pos.end(), pos.begin() + first, pos.begin() + n);
        if (sk_) sk_->insert(sk_->end(), sk.begin() + first, sk.begin() + n);
        return Output<CANONICAL>(SYNCMER ? k_ + w_ - 1 : k_, seq, &min_pos, &ws);
    }
    // src/lib.rs:553-576: `run` with the scratch passed explicitly; the reference's Cache is this
    // engine's Workspace (device buffers + stream)
    Output<CANONICAL> run_with_buf(PackedSeq seq, std::vector<uint32_t> &min_pos, Workspace &cache) const {
        return workspace(&cache).run(seq, min_pos);
    }
    // src/lib.rs:370-376, :517-543.  The reference's scalar collectors OVERWRITE min_pos from index 0
    // and truncate it to the result (src/collect.rs:15-37,39-76, src/syncmers.rs:19-48): no append, no
    // last() rule; a sequence without a window clears min_pos and leaves the super-k-mer vector
    // untouched (src/collect.rs:45-48).  Served by the same HIP kernel as `run`.
    Output<CANONICAL> run_scalar(PackedSeq seq, std::vector<uint32_t> &min_pos) const {
        Workspace &ws = ws_ ? *ws_ : Workspace::thread_default();
        std::vector<uint32_t> pos, sk;
        const uint64_t n = compute(seq, ws, pos, sk);
        min_pos.assign(pos.begin(), pos.begin() + n);
        if (sk_ && seq.len >= (uint64_t)k_ + w_ - 1) sk_->assign(sk.begin(), sk.begin() + n);
        return Output<CANONICAL>(SYNCMER ? k_ + w_ - 1 : k_, seq, &min_pos, &ws);
    }
    std::vector<uint32_t> run_scalar_once(PackedSeq seq) const {  // src/lib.rs:358-362, :511-515
        std::vector<uint32_t> v;
        run_scalar(seq, v);
        return v;
    }
    // src/lib.rs:451-496: canonical builders only; windows with an ambiguous base are skipped
    Output<CANONICAL> run_skip_ambiguous_windows(PackedNSeq nseq, std::vector<uint32_t> &min_pos) const {
        static_assert(CANONICAL, "run_skip_ambiguous_windows() is only defined for canonical builders");
        Workspace &ws = ws_ ? *ws_ : Workspace::thread_default();
        mm_plan_t *plan = nullptr;
        check(mm_plan_create(&plan, k_, w_, CANONICAL, (mm_mode_t)SYNCMER, has_hasher_ ? &hasher_ : nullptr));
        const uint64_t l = (uint64_t)k_ + w_ - 1;
        const uint64_t cap = nseq.seq.len >= l ? nseq.seq.len - l + 1 : 0;
        std::vector<uint32_t> pos(cap ? cap : 1);
        uint64_t n = 0;
        int r = mm_run_skip_ambiguous_host(plan, ws.get(), nseq.seq.data, nseq.seq.offset, nseq.amb,
                                           nseq.amb_offset, nseq.seq.len, pos.data(), cap, &n);
        mm_plan_destroy(plan);
        check(r);
        size_t first = 0;
        if (SYNCMER == 0)
            while (first < n && !min_pos.empty() && pos[first] == min_pos.back()) ++first;
        min_pos.insert(min_pos.end(), pos.begin() + first, pos.begin() + n);
        return Output<CANONICAL>(SYNCMER ? k_ + w_ - 1 : k_, nseq.seq, &min_pos, &ws);
    }
    Output<CANONICAL> run_skip_ambiguous_windows_with_buf(PackedNSeq nseq, std::vector<uint32_t> &min_pos,
                                                          Workspace &cache) const {  // src/lib.rs:465-496
        return workspace(&cache).run_skip_ambiguous_windows(nseq, min_pos);
    }
    std::vector<uint32_t> run_skip_ambiguous_windows_once(PackedNSeq nseq) const {
        std::vector<uint32_t> v;
        run_skip_ambiguous_windows(nseq, v);
        return v;
    }
    std::vector<uint32_t> run_once(PackedSeq seq) const {  // src/lib.rs:364
        std::vector<uint32_t> v;
        run(seq, v);
        return v;
    }
    std::vector<uint32_t> run_once(AsciiSeq seq) const {
        Workspace &ws = ws_ ? *ws_ : Workspace::thread_default();
        mm_plan_t *plan = nullptr;
        check(mm_plan_create(&plan, k_, w_, CANONICAL, (mm_mode_t)SYNCMER, has_hasher_ ? &hasher_ : nullptr));
        const uint64_t l = (uint64_t)k_ + w_ - 1;
        const uint64_t cap = seq.len >= l ? seq.len - l + 1 : 0;
        std::vector<uint32_t> pos(cap ? cap : 1);
        uint64_t n = 0;
        int r = mm_run_host_ascii(plan, ws.get(), seq.data, seq.len, pos.data(), nullptr, cap, &n);
        mm_plan_destroy(plan);
        check(r);
        pos.resize(n);
        return pos;
    }

    // MANY sequences in ONE call (round 6) - what a caller that loops `run` over its reads does instead (src/lib.rs:378
    // per read; the reference's `short` experiment, bench/src/bin/paper.rs:62-115): the reads are packed back to back on
    // the host, one upload, ONE launch whatever their lengths (a read longer than a lane takes consecutive lanes of the
    // device-built lane table), one download.  Positions are read-local; read r's are pos[offsets[r] .. offsets[r + 1]).
    // With .super_kmers(&sk) the indices come back the same way.  pos / offsets / sk are OVERWRITTEN.
    void run_many(const std::vector<PackedSeq> &reads, std::vector<uint32_t> &pos, std::vector<uint64_t> &offsets) const {
        Workspace &ws = ws_ ? *ws_ : Workspace::thread_default();
        std::vector<uint64_t> starts(reads.size() + 1, 0);
        uint32_t longest = 0;
        for (size_t r = 0; r < reads.size(); ++r) {
            starts[r + 1] = starts[r] + reads[r].len;
            if (reads[r].len > longest) longest = (uint32_t)reads[r].len;
        }
        const uint64_t total = starts.back();
        std::vector<uint8_t> packed((total + 3) / 4 + 16, 0);
        for (size_t r = 0; r < reads.size(); ++r)  // (base by base: any source offset to any destination offset)
            for (uint64_t i = 0; i < reads[r].len; ++i) {
                const uint64_t s = reads[r].offset + i, d = starts[r] + i;
                packed[d >> 2] |= (uint8_t)(((reads[r].data[s >> 2] >> (2 * (s & 3))) & 3u) << (2 * (d & 3)));
            }
        mm_plan_t *plan = nullptr;
        check(mm_plan_create(&plan, k_, w_, CANONICAL, (mm_mode_t)SYNCMER, has_hasher_ ? &hasher_ : nullptr));
        const uint64_t cap = total ? total : 1;
        pos.assign(cap, 0);
        std::vector<uint32_t> sk(sk_ ? cap : 0);
        offsets.assign(reads.size() + 1, 0);
        uint64_t n = 0;
        const int r = mm_run_packed_reads_host(plan, ws.get(), packed.data(), reads.size(), starts.data(), longest, pos.data(),
                                               sk_ ? sk.data() : nullptr, cap, offsets.data(), &n);
        mm_plan_destroy(plan);
        check(r);
        pos.resize(n);
        if (sk_) sk_->assign(sk.begin(), sk.begin() + n);
    }

    // the immutable plan of this builder (caller destroys it); k and w
    mm_plan_t *make_plan() const {
        mm_plan_t *plan = nullptr;
        check(mm_plan_create(&plan, k_, w_, CANONICAL, (mm_mode_t)SYNCMER, has_hasher_ ? &hasher_ : nullptr));
        return plan;
    }
    uint32_t k() const { return k_; }
    uint32_t w() const { return w_; }

  private:
    // one pass of the hot path over `seq` on the device; fills pos (and sk), returns the count
    uint64_t compute(PackedSeq seq, Workspace &ws, std::vector<uint32_t> &pos, std::vector<uint32_t> &sk) const {
        mm_plan_t *plan = nullptr;
        check(mm_plan_create(&plan, k_, w_, CANONICAL, (mm_mode_t)SYNCMER, has_hasher_ ? &hasher_ : nullptr));
        const uint64_t l = (uint64_t)k_ + w_ - 1;
        const uint64_t cap = seq.len >= l ? seq.len - l + 1 : 0;
        pos.assign(cap ? cap : 1, 0);
        sk.assign(sk_ ? (cap ? cap : 1) : 0, 0);
        uint64_t n = 0;
        const int r = mm_run_host(plan, ws.get(), seq.data, seq.offset, seq.len, pos.data(),
                                  sk_ ? sk.data() : nullptr, cap, &n);
        mm_plan_destroy(plan);
        check(r);
        return n;
    }

    uint32_t k_, w_;
    mm_hasher_t hasher_{};
    bool has_hasher_ = false;
    std::vector<uint32_t> *sk_ = nullptr;
    Workspace *ws_ = nullptr;
};

// Several devices from one call: the reference's rayon loop over contigs (bench/src/bin/paper.rs:442-459) behind
// the C ABI.  A group holds one workspace per listed device (a device may be listed more than once).
class DeviceGroup {
  public:
    explicit DeviceGroup(const std::vector<int> &devices) {
        check(mm_device_group_create(&g_, devices.data(), (int)devices.size()));
    }
    ~DeviceGroup() { mm_device_group_destroy(g_); }
    DeviceGroup(const DeviceGroup &) = delete;
    DeviceGroup &operator=(const DeviceGroup &) = delete;
    mm_device_group_t *get() const { return g_; }
    int size() const { return mm_device_group_size(g_); }

    // Builder::run over all devices of the group: one sequence cut into window ranges (absolute positions,
    // exact seam); positions are APPENDED to min_pos with the last() rule like run()
    template <bool CANONICAL, int SYNCMER>
    void run(const Builder<CANONICAL, SYNCMER> &b, PackedSeq seq, std::vector<uint32_t> &min_pos) const {
        mm_plan_t *plan = b.make_plan();
        const uint64_t l = (uint64_t)b.k() + b.w() - 1;
        const uint64_t cap = seq.len >= l ? seq.len - l + 1 : 0;
        std::vector<uint32_t> pos(cap ? cap : 1);
        uint64_t n = 0;
        const int r = mm_run_sharded_host(plan, g_, seq.data, seq.offset, seq.len, pos.data(), nullptr, cap, &n);
        mm_plan_destroy(plan);
        check(r);
        size_t first = 0;
        if (SYNCMER == 0)
            while (first < n && !min_pos.empty() && pos[first] == min_pos.back()) ++first;
        min_pos.insert(min_pos.end(), pos.begin() + first, pos.begin() + n);
    }
    // one Builder::run per sequence (paper.rs:425-431), the sequences placed greedily on the devices: positions
    // are sequence-local, sequence s = pos[offsets[s] .. offsets[s + 1])
    template <bool CANONICAL, int SYNCMER>
    void run_batch(const Builder<CANONICAL, SYNCMER> &b, const std::vector<PackedSeq> &seqs, std::vector<uint32_t> &pos,
                   std::vector<uint64_t> &offsets) const {
        mm_plan_t *plan = b.make_plan();
        std::vector<const uint8_t *> ptr;
        std::vector<uint64_t> off, len;
        uint64_t cap = 1;
        for (const PackedSeq &s : seqs) {
            ptr.push_back(s.data);
            off.push_back(s.offset);
            len.push_back(s.len);
            cap += s.len;
        }
        pos.assign(cap, 0);
        offsets.assign(seqs.size() + 1, 0);
        const int r = mm_run_batch_sharded_host(plan, g_, seqs.size(), ptr.data(), off.data(), len.data(), pos.data(),
                                                nullptr, cap, offsets.data());
        mm_plan_destroy(plan);
        check(r);
        pos.resize(offsets.back());
    }

    // ---- device-resident shards (round 4): the sequence lives in HBM on every device of the group, every device
    // walks its window range with one asynchronous launch, the positions stay where they were made
    void upload(PackedSeq seq) { check(mm_device_group_upload(g_, seq.data, (seq.offset + seq.len + 3) / 4)); }
    /// Every device receives only what its share of an N-way split of `seq` reads (one crossing of the host link in total).
    void upload_shares(PackedSeq seq) {
        check(mm_device_group_upload_range(g_, seq.data, (seq.offset + seq.len + 3) / 4, seq.offset, seq.len));
    }
    void adopt(const std::vector<const void *> &d_packed, uint64_t packed_bytes) {
        check(mm_device_group_adopt(g_, d_packed.data(), packed_bytes));
    }
    // per-entry counts of Builder::run over the resident sequence (n_bases bases from base_offset on)
    template <bool CANONICAL, int SYNCMER>
    std::vector<uint64_t> run_device(const Builder<CANONICAL, SYNCMER> &b, uint64_t n_bases, uint64_t base_offset = 0,
                                     bool super_kmers = false) const {
        mm_plan_t *plan = b.make_plan();
        std::vector<uint64_t> counts((size_t)size());
        const int r = mm_run_sharded_device(plan, g_, base_offset, n_bases, super_kmers ? 1 : 0, counts.data(), nullptr);
        mm_plan_destroy(plan);
        check(r);
        return counts;
    }
    struct Shard {
        uint32_t *d_pos = nullptr, *d_sk = nullptr;  // device pointers on the entry's device
        uint64_t count = 0, win_begin = 0, win_end = 0;
    };
    Shard result(int entry) const {
        Shard s;
        check(mm_device_group_result(g_, entry, &s.d_pos, &s.d_sk, &s.count, &s.win_begin, &s.win_end));
        return s;
    }
    // device-resident BATCHES: independent sequences placed greedily on the entries, each on its entry's device only;
    // one batch launch per entry; sequence-local positions stay on the devices (mm_device_group_batch_result)
    void upload_batch(const std::vector<PackedSeq> &seqs) {
        std::vector<const uint8_t *> ptr;
        std::vector<uint64_t> bytes;
        for (const PackedSeq &s : seqs) {
            ptr.push_back(s.data);
            bytes.push_back((s.offset + s.len + 3) / 4);
        }
        check(mm_device_group_upload_batch(g_, seqs.size(), ptr.data(), bytes.data()));
    }
    template <bool CANONICAL, int SYNCMER>
    std::vector<uint64_t> run_batch_device(const Builder<CANONICAL, SYNCMER> &b, const std::vector<PackedSeq> &seqs) const {
        mm_plan_t *plan = b.make_plan();
        std::vector<uint64_t> off, len, counts(seqs.size());
        for (const PackedSeq &s : seqs) {
            off.push_back(s.offset);
            len.push_back(s.len);
        }
        const int r = mm_run_batch_sharded_device(plan, g_, off.data(), len.data(), 0, counts.data(), nullptr);
        mm_plan_destroy(plan);
        check(r);
        return counts;
    }
    // all sequences, input order, into device memory of entry `root`; returns the n + 1 offsets
    std::vector<uint64_t> gather_batch(int root, size_t n_seqs, uint32_t *d_dst_pos, uint64_t capacity) const {
        std::vector<uint64_t> offs(n_seqs + 1);
        check(mm_device_group_gather_batch(g_, root, d_dst_pos, nullptr, capacity, offs.data()));
        return offs;
    }
    // the shards, dense and in window order, into device memory of entry `root` (device-to-device copies)
    uint64_t gather(int root, uint32_t *d_dst_pos, uint64_t capacity, uint32_t *d_dst_sk = nullptr) const {
        uint64_t total = 0;
        check(mm_device_group_gather(g_, root, d_dst_pos, d_dst_sk, capacity, &total));
        return total;
    }

  private:
    mm_device_group_t *g_ = nullptr;
};

// constructors, src/lib.rs:240-321
inline Builder<false, 0> minimizers(uint32_t k, uint32_t w) { return {k, w}; }
inline Builder<true, 0> canonical_minimizers(uint32_t k, uint32_t w) { return {k, w}; }
inline Builder<false, 1> closed_syncmers(uint32_t k, uint32_t w) { return {k, w}; }
inline Builder<true, 1> canonical_closed_syncmers(uint32_t k, uint32_t w) { return {k, w}; }
inline Builder<false, 2> open_syncmers(uint32_t k, uint32_t w) { return {k, w}; }
inline Builder<true, 2> canonical_open_syncmers(uint32_t k, uint32_t w) { return {k, w}; }
inline Builder<true, 1> canonical_syncmers(uint32_t k, uint32_t w) { return {k, w}; }  // README.md:65 name

// free functions, src/lib.rs:639-654
inline std::vector<uint32_t> minimizer_positions(PackedSeq s, uint32_t k, uint32_t w) { return minimizers(k, w).run_once(s); }
inline std::vector<uint32_t> minimizer_positions(AsciiSeq s, uint32_t k, uint32_t w) { return minimizers(k, w).run_once(s); }
inline std::vector<uint32_t> canonical_minimizer_positions(PackedSeq s, uint32_t k, uint32_t w) {
    return canonical_minimizers(k, w).run_once(s);
}

}  // namespace simd_minimizers
