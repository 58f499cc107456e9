// Drives the C++ AlacContext mirror the way the reference's callers do (ALACFileReader.cs:89-116, Program.cs:39-49):
//   open -> GetNumSamples / format getters -> while ((n = Read(buf)) > 0) ...   and   SetPosition + Read.
// usage: alaccontext_selftest file.m4a [seek_position_in_samples]
//        alaccontext_selftest file.m4a reader <chunk_bytes> [seek_position_in_bytes]    the NAudio adapter's Read(buffer, offset, count) loop
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include <cstring>

#include "ALACFileReader.hpp"

static unsigned long long fnv(unsigned long long h, const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s file.m4a [seek]\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) return 2;
    try {
        if (argc >= 4 && std::strcmp(argv[2], "reader") == 0) {
            AlacNetNAudioAdapter::ALACFileReader reader(f, 0, 5);
            const int chunk = std::atoi(argv[3]);
            std::vector<uint8_t> rb((size_t)chunk + 16);
            if (argc >= 5) {
                reader.Read(rb.data(), 0, chunk);
                reader.SetPosition(std::atoll(argv[4]));
            }
            unsigned long long h = 1469598103934665603ull;
            long long total = 0;
            for (;;) {
                int n = reader.Read(rb.data(), 3, chunk);          // (offset 3: the adapter honours it)
                if (n <= 0) break;
                h = fnv(h, rb.data() + 3, (size_t)n);
                total += n;
            }
            const auto& wf = reader.GetWaveFormat();
            std::printf("reader rate=%d bits=%d channels=%d align=%d length=%lld bytes=%lld fnv=%llu position=%lld\n", wf.SampleRate,
                        wf.BitsPerSample, wf.Channels, wf.BlockAlign, reader.Length, total, h, reader.GetPosition());
            return 0;
        }
        ALACdotNET::Decoder::AlacContext ctx(f, 0, 7);
        std::vector<uint8_t> buf(1024 * 80);
        if (argc >= 3) {
            ctx.Read(buf.data());
            ctx.SetPosition(std::atoll(argv[2]));
            int n = ctx.Read(buf.data());
            std::printf("seek bytes=%d fnv=%llu last=%d\n", n, fnv(1469598103934665603ull, buf.data(), (size_t)(n > 0 ? n : 0)), ctx.LastSampleNumber);
            return 0;
        }
        unsigned long long h = 1469598103934665603ull;
        long long total = 0;
        for (;;) {
            int n = ctx.Read(buf.data());
            if (n <= 0) break;
            h = fnv(h, buf.data(), (size_t)n);
            total += n;
        }
        std::printf("rate=%d channels=%d bits=%d samples=%d bytes=%lld fnv=%llu last=%d\n", ctx.GetSampleRate(), ctx.GetNumChannels(),
                    ctx.GetBitsPerSample(), ctx.GetNumSamples(), total, h, ctx.LastSampleNumber);
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
