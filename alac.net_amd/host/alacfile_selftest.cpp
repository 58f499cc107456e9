// Exercises the C++ host mirror the way AlacContext.cs:54-55,:197 drives the reference:
//   new AlacFile(sampleSize, numChannels); SetInfo(codecData); DecodeFrame(readBuffer, destBuffer)
// Reads a packet file (argv[1]) written by the tests, decodes it, prints the return value and a checksum.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "AlacFile.hpp"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s packet.bin samplesize numchannels\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> pkt(1024 * 80);
    size_t n = std::fread(pkt.data(), 1, pkt.size(), f);
    std::fclose(f);
    const int ss = std::atoi(argv[2]), nc = std::atoi(argv[3]);
    int32_t cd[48] = {0};
    cd[26] = 0x10; /* 4096 samples/frame */
    cd[29] = ss; cd[30] = 40; cd[31] = 10; cd[32] = 14; cd[33] = nc;
    try {
        ALACdotNET::Decoder::AlacFile alac(ss, nc);
        alac.SetInfo(cd);
        std::vector<int32_t> out(1024 * 80);
        int bytes = alac.DecodeFrame(pkt.data(), (uint32_t)n, out.data());
        unsigned long long sum = 1469598103934665603ull;
        const int ints = bytes / (ss / 8) * (ss == 24 ? 3 : 1);
        for (int i = 0; i < ints; i++) { sum ^= (uint32_t)out[i]; sum *= 1099511628211ull; }
        std::printf("bytes=%d ints=%d fnv=%llu\n", bytes, ints, sum);
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
