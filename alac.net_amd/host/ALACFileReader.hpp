// ALACFileReader.hpp -- C++ twin of the reference's NAudio adapter (AlacNetNAudioAdapter/ALACFileReader.cs:22-126) over the
// GPU-backed AlacContext of AlacContext.hpp: WaveFormat, Length, Position get/set, Read(buffer, offset, count), one lock.
// NAudio (WaveStream / WaveFormat) is a Windows audio library outside this path; WaveFormat carries the fields the
// adapter and its callers use.  Executed by alaccontext_selftest (`reader` mode) and tests/test_container.py.
#pragma once
#include <algorithm>
#include <mutex>
#include <vector>

#include "AlacContext.hpp"

namespace AlacNetNAudioAdapter {

struct WaveFormat {   // new WaveFormat(rate, bits, channels)
    int SampleRate = 0, BitsPerSample = 0, Channels = 0, BlockAlign = 0, AverageBytesPerSecond = 0;
    WaveFormat() = default;
    WaveFormat(int rate, int bits, int channels)
        : SampleRate(rate), BitsPerSample(bits), Channels(channels), BlockAlign(channels * (bits / 8)),
          AverageBytesPerSecond(rate * channels * (bits / 8)) {}
};

class ALACFileReader {
public:
    explicit ALACFileReader(std::istream& baseStream, int device = 0, int batchPackets = 256)
        : alacContext_(baseStream, device, batchPackets),                                                   // :41
          waveFormat_(alacContext_.GetSampleRate(), alacContext_.GetBytesPerSample() * 8, alacContext_.GetNumChannels()),   // :42
          decompressBuffer_((size_t)65546 * (size_t)(waveFormat_.BitsPerSample / 8) * (size_t)waveFormat_.Channels) {   // :44
        Length = (long long)alacContext_.GetNumSamples() * waveFormat_.BlockAlign;                           // :43
    }

    const WaveFormat& GetWaveFormat() const { return waveFormat_; }
    long long Length = 0;   // bytes of the uncompressed wave stream

    long long GetPosition() { return (long long)alacContext_.LastSampleNumber * waveFormat_.BlockAlign; }   // :65
    void SetPosition(long long value) {                                                                      // :66-73
        std::lock_guard<std::mutex> g(repositionLock_);
        alacContext_.SetPosition(value / waveFormat_.BlockAlign);
        decompressLeftovers_ = 0;   // after repositioning no more data comes from the buffer
    }

    int Read(uint8_t* buffer, int offset, int count) {                                                       // :89-116
        int bytesRead = 0;
        std::lock_guard<std::mutex> g(repositionLock_);
        while (bytesRead < count) {
            if (decompressLeftovers_ > 0) {
                const int toCopy = std::min(decompressLeftovers_, count - bytesRead);
                std::memcpy(buffer + offset, decompressBuffer_.data() + decompressBufferOffset_, (size_t)toCopy);
                decompressLeftovers_ -= toCopy;
                decompressBufferOffset_ = decompressLeftovers_ == 0 ? 0 : decompressBufferOffset_ + toCopy;
                bytesRead += toCopy;
                offset += toCopy;
            }
            if (bytesRead >= count) break;
            decompressBufferOffset_ = 0;
            const int bytesUnpacked = alacContext_.Read(decompressBuffer_.data());
            if (bytesUnpacked == 0) break;
            decompressLeftovers_ += bytesUnpacked;
        }
        return bytesRead;
    }

private:
    ALACdotNET::Decoder::AlacContext alacContext_;
    WaveFormat waveFormat_;
    std::vector<uint8_t> decompressBuffer_;
    int decompressLeftovers_ = 0, decompressBufferOffset_ = 0;
    std::mutex repositionLock_;
};

}  // namespace AlacNetNAudioAdapter
