// Reads like the reference's doctests (src/lib.rs:87-141) through the C++ mirror of the builder.
// Exit code 0 = all known answers reproduced; 77 = no GPU (the engine has no CPU fallback).
#include <cstdio>
#include <cstring>
#include <string>

#include "simd_minimizers_amd.hpp"

using namespace simd_minimizers;

static std::vector<uint8_t> pack(const char *s) {
    size_t n = strlen(s);
    std::vector<uint8_t> p((n + 3) / 4 + 16, 0);
    for (size_t i = 0; i < n; ++i) p[i / 4] |= (uint8_t)(((s[i] >> 1) & 3) << (2 * (i % 4)));
    return p;
}

int main() {
    if (mm_device_count() <= 0) {
        try {
            const char *seq = "ACGTGCTCAGAGACTCAG";
            minimizer_positions(AsciiSeq{(const uint8_t *)seq, strlen(seq)}, 5, 7);
        } catch (const Error &e) {
            printf("no GPU: %s (code %d)\n", e.what(), e.code);
            return e.code == MM_ERR_NO_DEVICE ? 77 : 1;
        }
        return 1;
    }
    const char *seq = "ACGTGCTCAGAGACTCAG";
    auto p1 = minimizer_positions(AsciiSeq{(const uint8_t *)seq, strlen(seq)}, 5, 7);
    if (p1 != std::vector<uint32_t>{4, 5, 8, 13}) return 2;

    const char *seq2 = "ACGTGCTCAGAGACTCAGAGGA";
    auto packed = pack(seq2);
    std::vector<uint32_t> pos, sk;
    auto out = canonical_minimizers(5, 7).super_kmers(&sk).run(PackedSeq{packed.data(), 0, strlen(seq2)}, pos);
    if (pos != std::vector<uint32_t>{0, 7, 9, 15}) return 3;
    if (out.values_u64() != std::vector<uint64_t>{0b1011010001, 0b1100110001, 0b0100110011, 0b1100110001}) return 4;
    if (sk.size() != pos.size()) return 5;
    try {
        canonical_minimizers(5, 6).run_once(PackedSeq{packed.data(), 0, strlen(seq2)});
        return 6;
    } catch (const Error &e) {
        if (e.code != MM_ERR_EVEN_L) return 7;
    }
    // skip-ambiguous windows (src/test.rs:428-482): no k-mer at an output position holds an N
    {
        const char *nseq_ascii = "ACGTGCTCAGNGACTCAGAGGATTACAGCTAGCTANCGATCGATTTAGC";
        const size_t n = strlen(nseq_ascii);
        auto pk = pack(nseq_ascii);
        std::vector<uint8_t> amb((n + 7) / 8 + 16, 0);
        for (size_t i = 0; i < n; ++i)
            if (nseq_ascii[i] == 'N') amb[i / 8] |= (uint8_t)(1u << (i % 8));
        auto p = canonical_minimizers(5, 7).run_skip_ambiguous_windows_once(
            PackedNSeq{PackedSeq{pk.data(), 0, n}, amb.data(), 0});
        if (p.empty()) return 8;
        for (uint32_t x : p)
            for (uint32_t j = 0; j < 5; ++j)
                if (nseq_ascii[x + j] == 'N') return 9;
    }
    // scalar flavour (src/lib.rs:358-376): the collectors OVERWRITE the output vector (src/collect.rs:24-25,36)
    {
        std::vector<uint32_t> v{111, 222, 333, 444, 555, 666};
        canonical_minimizers(5, 7).run_scalar(PackedSeq{packed.data(), 0, strlen(seq2)}, v);
        if (v != std::vector<uint32_t>{0, 7, 9, 15}) return 10;
        if (canonical_minimizers(5, 7).run_scalar_once(PackedSeq{packed.data(), 0, strlen(seq2)}) != v) return 11;
        // SIMD flavour appends and drops a leading result equal to last() (src/collect.rs:265-271)
        std::vector<uint32_t> a{0};
        canonical_minimizers(5, 7).run(PackedSeq{packed.data(), 0, strlen(seq2)}, a);
        if (a != std::vector<uint32_t>{0, 7, 9, 15}) return 12;
        // too short for a window: scalar clears min_pos and leaves the super-k-mer vector alone
        std::vector<uint32_t> m{1, 2, 3}, s2{9, 9};
        canonical_minimizers(5, 7).super_kmers(&s2).run_scalar(PackedSeq{packed.data(), 0, 5}, m);
        if (!m.empty() || s2 != std::vector<uint32_t>{9, 9}) return 13;
        // explicit scratch (src/lib.rs:553-559): Cache == Workspace
        Workspace cache(0);
        std::vector<uint32_t> b;
        auto o2 = canonical_minimizers(5, 7).run_with_buf(PackedSeq{packed.data(), 0, strlen(seq2)}, b, cache);
        if (b != std::vector<uint32_t>{0, 7, 9, 15}) return 14;
        // Output::values_u128 / pos_and_values_u64 / pos_and_values_u128 (src/lib.rs:587-630)
        auto v128 = o2.values_u128();
        auto pv64 = o2.pos_and_values_u64();
        auto pv128 = o2.pos_and_values_u128();
        const uint64_t want[4] = {0b1011010001, 0b1100110001, 0b0100110011, 0b1100110001};
        if (v128.size() != 4 || pv64.size() != 4 || pv128.size() != 4) return 15;
        for (int i = 0; i < 4; ++i) {
            if (v128[i] != (u128)want[i]) return 16;
            if (pv64[i].first != b[i] || pv64[i].second != want[i]) return 17;
            if (pv128[i].first != b[i] || pv128[i].second != (u128)want[i]) return 18;
        }
        // l-mer values wider than 64 bits: closed syncmers k=21 w=21 (l = 41) on all-G (src/test.rs:576-597 shape)
        std::string g(60, 'G');
        auto pg = pack(g.c_str());
        std::vector<uint32_t> sp;
        auto og = closed_syncmers(21, 21).run(PackedSeq{pg.data(), 0, g.size()}, sp);
        auto gv = og.values_u128();
        if (gv.empty()) return 19;
        const u128 allg = (((u128)1 << 82) - 1);  // 41 bases of code 3
        for (u128 x : gv)
            if (x != allg) return 20;
    }
    // several devices from one call (bench/src/bin/paper.rs:442-459): two workspaces on device 0
    {
        DeviceGroup group({0, 0});
        if (group.size() != 2) return 21;
        std::string big;
        uint64_t z = 12345;
        for (int i = 0; i < 200000; ++i) {
            z = z * 6364136223846793005ull + 1442695040888963407ull;
            big.push_back("ACGT"[(z >> 33) & 3]);
        }
        auto pb = pack(big.c_str());
        const PackedSeq whole{pb.data(), 0, big.size()};
        std::vector<uint32_t> one, two;
        canonical_minimizers(21, 11).run(whole, one);
        group.run(canonical_minimizers(21, 11), whole, two);
        if (one.empty() || one != two) return 22;
        // contigs: slices of the same sequence, sequence-local positions in input order
        std::vector<PackedSeq> parts{PackedSeq{pb.data(), 0, 50000}, PackedSeq{pb.data() + 12500, 1, 70001},
                                     PackedSeq{pb.data(), 0, 10}, PackedSeq{pb.data() + 30000, 2, 79000}};
        std::vector<uint32_t> pos;
        std::vector<uint64_t> offs;
        group.run_batch(canonical_minimizers(21, 11), parts, pos, offs);
        if (offs.size() != 5) return 23;
        for (size_t i = 0; i < parts.size(); ++i) {
            std::vector<uint32_t> want = canonical_minimizers(21, 11).run_once(parts[i]);
            if (std::vector<uint32_t>(pos.begin() + offs[i], pos.begin() + offs[i + 1]) != want) return 24 + (int)i;
        }
        // device-resident shards (round 4): the sequence uploaded once, one launch per entry, the counts back; the
        // shards' window ranges are contiguous and their counts add up to the single-device result
        group.upload(whole);
        const std::vector<uint64_t> counts = group.run_device(canonical_minimizers(21, 11), whole.len);
        uint64_t sum = 0, prev_end = 0;
        for (int i = 0; i < group.size(); ++i) {
            const DeviceGroup::Shard sh = group.result(i);
            if (sh.count != counts[(size_t)i] || sh.win_begin != prev_end || (sh.count && !sh.d_pos)) return 30;
            prev_end = sh.win_end;
            sum += sh.count;
        }
        if (sum != one.size() || prev_end != big.size() - 31 + 1) return 31;
        // device-resident batches: the contigs above, each on one entry's device; counts per contig
        group.upload_batch(parts);
        const std::vector<uint64_t> cc = group.run_batch_device(canonical_minimizers(21, 11), parts);
        for (size_t i = 0; i < parts.size(); ++i)
            if (cc[i] != offs[i + 1] - offs[i]) return 32;
    }
    // round 6: many reads of ANY lengths in one call (Builder::run_many -> mm_run_packed_reads_host -> one lane-table
    // launch): every read equals its own run_once, super-k-mer indices too, at odd source offsets
    {
        std::string big;
        uint64_t x = 99;
        for (int i = 0; i < 120000; ++i) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            big.push_back("ACGT"[(x >> 33) & 3]);
        }
        auto pb = pack(big.c_str());
        const std::vector<PackedSeq> reads = {PackedSeq{pb.data(), 0, 150},     PackedSeq{pb.data() + 50, 1, 40001}, PackedSeq{pb.data(), 3, 0},
                                              PackedSeq{pb.data() + 7, 2, 30},  PackedSeq{pb.data() + 9000, 3, 7003}, PackedSeq{pb.data() + 20000, 0, 31},
                                              PackedSeq{pb.data() + 100, 2, 900}};
        std::vector<uint32_t> pos, sk;
        std::vector<uint64_t> offs;
        canonical_minimizers(21, 11).super_kmers(&sk).run_many(reads, pos, offs);
        if (offs.size() != reads.size() + 1 || offs.back() != pos.size() || sk.size() != pos.size()) return 40;
        for (size_t i = 0; i < reads.size(); ++i) {
            std::vector<uint32_t> wsk;
            std::vector<uint32_t> want;
            canonical_minimizers(21, 11).super_kmers(&wsk).run(reads[i], want);
            if (std::vector<uint32_t>(pos.begin() + offs[i], pos.begin() + offs[i + 1]) != want) return 41 + (int)i;
            if (std::vector<uint32_t>(sk.begin() + offs[i], sk.begin() + offs[i + 1]) != wsk) return 51 + (int)i;
        }
        std::vector<uint32_t> p2;
        closed_syncmers(15, 17).run_many(reads, p2, offs);
        for (size_t i = 0; i < reads.size(); ++i)
            if (std::vector<uint32_t>(p2.begin() + offs[i], p2.begin() + offs[i + 1]) != closed_syncmers(15, 17).run_once(reads[i])) return 61 + (int)i;
    }
    printf("builder_example ok\n");
    return 0;
}
