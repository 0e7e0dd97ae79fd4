// Reads like the reference's doctests (src/lib.rs:87-141) through the C++ mirror of the builder.
// Exit code 0 = all known answers reproduced; 77 = no GPU (the engine has no CPU fallback).
#include <cstdio>
#include <cstring>

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
    printf("builder_example ok\n");
    return 0;
}
