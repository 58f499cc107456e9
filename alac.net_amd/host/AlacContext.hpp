// AlacContext.hpp -- C++ host-side mirror of the reference's container/session layer on top of the C ABI:
//   * QtMovieT::ReadHeader  (ALACDecoder/QTMovieT.cs:51-752)  MP4/M4A atoms -> DemuxResT (DemuxResT.cs:22-34)
//   * AlacContext           (ALACDecoder/AlacContext.cs:20-338) public surface: getters, Read(buffer),
//                           SetPosition(position), LastSampleNumber -- plus the batch entry point ReadBatch.
// Same acceptance rules and quirks as the reference (see alac.net_amd/container.py, the Python twin, for the list):
// moov must precede mdat (QTMovieT.cs:746), <= 16 stts entries, smhd(16)/dinf/stbl order inside minf, the post-seek
// int offset (AlacContext.cs:200-202) and the LastSampleNumber double count after a seek (:199,:283).
// Read() keeps the reference's contract (one packet per call, little-endian PCM) but serves from batches decoded on
// the GPU: the next K packets are pre-read with the sizes the demuxer holds and submitted in ONE alacgpu_decode_batch.
#pragma once
#include <cstdint>
#include <cstring>
#include <deque>
#include <istream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "alacgpu.h"

namespace ALACdotNET { namespace Decoder {

enum class MdatPosStatus { None = 0, Ok = 1, NoValidSaveMdatPosition = 2, CannotSeekToMdatPosition = 3 };

struct DemuxResT {   // DemuxResT.cs:22-34
    int FormatRead = 0, NumChannels = 0, SampleSize = 0, SampleRate = 0;
    uint32_t Format = 0;
    std::vector<std::pair<int, int>> TimeToSample;   // (SampleCount, SampleDuration), at most 16
    int NumTimeToSamples = 0;
    std::vector<int> SampleByteSize;
    int CodecDataLength = 0;
    std::vector<int32_t> CodecData = std::vector<int32_t>(1024, 0);
    std::vector<int> Stco;
    struct ChunkInfo { int FirstChunk, SamplesPerChunk, SampleDescriptionIndex; };
    std::vector<ChunkInfo> Stsc;
    int MdatLen = 0;
};

class MyStream {   // MyStream.cs:14-115
public:
    explicit MyStream(std::istream& s) : s_(s) {
        auto pos = s_.tellg();
        s_.seekg(0, std::ios::end);
        length_ = (long long)s_.tellg();
        s_.seekg(pos);
    }
    bool EOFReached() { return Position() >= length_; }
    long long Position() { s_.clear(); return (long long)s_.tellg(); }
    int32_t ReadUint32() { uint8_t b[4]; Read(4, b); return (int32_t)(((uint32_t)b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3]); }
    int ReadUint16() { uint8_t b[2]; Read(2, b); return (b[0] << 8) | b[1]; }
    int ReadUint8() { uint8_t b; Read(1, &b); return b; }
    void Read(size_t n, uint8_t* dst) {
        std::memset(dst, 0, n);
        s_.clear();
        s_.read(reinterpret_cast<char*>(dst), (std::streamsize)n);
        s_.clear();
    }
    void Skip(long long n) { s_.clear(); s_.seekg(n, std::ios::cur); }
    long long Seek(long long pos) { s_.clear(); s_.seekg(pos, std::ios::beg); return Position(); }
private:
    std::istream& s_;
    long long length_ = 0;
};

inline uint32_t FourCc(const char* s) { return ((uint32_t)(uint8_t)s[0] << 24) | ((uint8_t)s[1] << 16) | ((uint8_t)s[2] << 8) | (uint8_t)s[3]; }

class QtMovieT {
public:
    QtMovieT(MyStream& s, DemuxResT& r) : s_(s), r_(r) {}

    MdatPosStatus ReadHeader() {   // QTMovieT.cs:51-109
        int foundMoov = 0, foundMdat = 0;
        for (;;) {
            int32_t len = s_.ReadUint32();
            if (s_.EOFReached()) return MdatPosStatus::None;
            uint32_t id = (uint32_t)s_.ReadUint32();
            if (id == FourCc("ftyp")) Ftyp(len);
            else if (id == FourCc("moov")) {
                if (!Container(len, 0)) return MdatPosStatus::None;
                if (foundMdat) return SetSavedMdat();
                foundMoov = 1;
            } else if (id == FourCc("mdat")) {
                Mdat(len, foundMoov ? 0 : 1);
                if (foundMoov) return MdatPosStatus::Ok;
                foundMdat = 1;
            } else if (id == FourCc("free") || id == FourCc("junk")) s_.Skip(len - 8);
            else return MdatPosStatus::None;
        }
    }

private:
    void Ftyp(int len) {   // :111-132
        int remaining = len - 8;
        uint32_t type = (uint32_t)s_.ReadUint32();
        remaining -= 4;
        if (type != FourCc("M4A ")) return;
        s_.ReadUint32();
        remaining -= 4;
        while (remaining != 0) { s_.ReadUint32(); remaining -= 4; }
    }
    // kind: 0 moov, 1 trak, 2 mdia, 3 stbl  (ReadChunkMoov/Trak/Media/Stbl)
    int Container(int len, int kind) {
        int remaining = len - 8;
        while (remaining != 0) {
            int32_t sub = s_.ReadUint32();
            if (sub <= 1 || sub > remaining) return 0;
            uint32_t id = (uint32_t)s_.ReadUint32();
            bool handled = true;
            if (kind == 0) {
                if (id == FourCc("mvhd") || id == FourCc("udta") || id == FourCc("elst") || id == FourCc("iods") || id == FourCc("free")) s_.Skip(sub - 8);
                else if (id == FourCc("trak")) { if (!Container(sub, 1)) return 0; }
                else handled = false;
            } else if (kind == 1) {
                if (id == FourCc("tkhd") || id == FourCc("edts")) s_.Skip(sub - 8);
                else if (id == FourCc("mdia")) { if (!Container(sub, 2)) return 0; }
                else handled = false;
            } else if (kind == 2) {
                if (id == FourCc("mdhd") || id == FourCc("hdlr")) s_.Skip(sub - 8);
                else if (id == FourCc("minf")) { if (!Minf(sub)) return 0; }
                else handled = false;
            } else {
                if (id == FourCc("stsd")) { if (!Stsd()) return 0; }
                else if (id == FourCc("stts")) Stts(sub);
                else if (id == FourCc("stsz")) Stsz(sub);
                else if (id == FourCc("stsc")) {
                    s_.Skip(4);
                    int n = s_.ReadUint32();
                    r_.Stsc.clear();
                    for (int i = 0; i < n; i++) { int a = s_.ReadUint32(), b = s_.ReadUint32(), c = s_.ReadUint32(); r_.Stsc.push_back({a, b, c}); }
                } else if (id == FourCc("stco")) {
                    s_.Skip(4);
                    int n = s_.ReadUint32();
                    r_.Stco.clear();
                    for (int i = 0; i < n; i++) r_.Stco.push_back(s_.ReadUint32());
                } else handled = false;
            }
            if (!handled) return 0;
            remaining -= sub;
        }
        return 1;
    }
    int Minf(int len) {   // :258-331
        int remaining = len - 8;
        if (s_.ReadUint32() != 16) return 0;
        if ((uint32_t)s_.ReadUint32() != FourCc("smhd")) return 0;
        s_.Skip(8);
        remaining -= 16;
        int dinf = s_.ReadUint32();
        if ((uint32_t)s_.ReadUint32() != FourCc("dinf")) return 0;
        s_.Skip(dinf - 8);
        remaining -= dinf;
        int stbl = s_.ReadUint32();
        if ((uint32_t)s_.ReadUint32() != FourCc("stbl")) return 0;
        if (!Container(stbl, 3)) return 0;
        remaining -= stbl;
        if (remaining != 0) s_.Skip(remaining);
        return 1;
    }
    int Stsd() {   // :412-523
        s_.Skip(4);
        if (s_.ReadUint32() != 1) return 0;
        int entrySize = s_.ReadUint32();
        r_.Format = (uint32_t)s_.ReadUint32();
        int remaining = entrySize - 8;
        if (r_.Format != FourCc("alac")) return 0;
        s_.Skip(6);
        s_.ReadUint16(); s_.ReadUint16(); s_.ReadUint32(); s_.ReadUint16();
        s_.Skip(4);
        s_.ReadUint16(); s_.ReadUint16();
        s_.Skip(4);
        remaining -= 6 + 2 + 6 + 2 + 4 + 4 + 4;
        r_.CodecDataLength = remaining + 12 + 8;
        if (r_.CodecDataLength > (int)r_.CodecData.size() || remaining < 0) return 0;
        for (int i = 0; i < r_.CodecDataLength; i++) r_.CodecData[i] = 0;
        r_.CodecData[0] = 0x0c000000;
        r_.CodecData[1] = (int32_t)FourCc("amrf");
        r_.CodecData[2] = (int32_t)FourCc("cala");
        std::vector<uint8_t> payload((size_t)remaining);
        s_.Read(payload.size(), payload.data());
        for (int i = 0; i < remaining; i++) r_.CodecData[12 + i] = payload[(size_t)i];
        r_.SampleSize = r_.CodecData[29] & 0xff;
        r_.NumChannels = r_.CodecData[33] & 0xff;
        r_.SampleRate = ((r_.CodecData[44] & 0xff) << 24) | ((r_.CodecData[45] & 0xff) << 16) | ((r_.CodecData[46] & 0xff) << 8) | (r_.CodecData[47] & 0xff);
        r_.FormatRead = 1;
        return 1;
    }
    void Stts(int len) {   // :525-559
        int remaining = len - 8;
        s_.Skip(4);
        int n = s_.ReadUint32();
        remaining -= 8;
        if (n > 16) throw std::out_of_range("Index was outside the bounds of the array.");   // DemuxResT.cs:27
        r_.NumTimeToSamples = n;
        r_.TimeToSample.clear();
        for (int i = 0; i < n; i++) { int c = s_.ReadUint32(), d = s_.ReadUint32(); r_.TimeToSample.push_back({c, d}); remaining -= 8; }
        if (remaining != 0) s_.Skip(remaining);
    }
    void Stsz(int len) {   // :561-613
        int remaining = len - 8;
        s_.Skip(4);
        int uniform = s_.ReadUint32();
        if (uniform != 0) { int n = s_.ReadUint32(); r_.SampleByteSize.assign((size_t)n, uniform); return; }
        int n = s_.ReadUint32();
        remaining -= 12;
        r_.SampleByteSize.resize((size_t)n);
        for (int i = 0; i < n; i++) { r_.SampleByteSize[(size_t)i] = s_.ReadUint32(); remaining -= 4; }
        if (remaining != 0) s_.Skip(remaining);
    }
    void Mdat(int len, int skip) {   // :724-734
        int remaining = len - 8;
        if (remaining == 0) return;
        r_.MdatLen = remaining;
        if (skip) { savedMdatPos_ = s_.Position(); s_.Skip(remaining); }
    }
    MdatPosStatus SetSavedMdat() {   // :736-750
        if (savedMdatPos_ == -1) return MdatPosStatus::NoValidSaveMdatPosition;
        if (s_.Seek(savedMdatPos_) != 0) return MdatPosStatus::CannotSeekToMdatPosition;
        return MdatPosStatus::Ok;
    }
    MyStream& s_;
    DemuxResT& r_;
    long long savedMdatPos_ = -1;
};

class AlacContext {
public:
    explicit AlacContext(std::istream& baseStream, int device = 0, int batchPackets = 256)
        : stream_(baseStream), batchPackets_(batchPackets < 1 ? 1 : batchPackets) {
        QtMovieT qt(stream_, demux_);
        MdatPosStatus st = qt.ReadHeader();
        if (st == MdatPosStatus::None || st == MdatPosStatus::CannotSeekToMdatPosition)
            throw std::runtime_error("Error while loading the QuickTime movie headers.");   // AlacContext.cs:50
        if (alacgpu_cfg_from_codec_data(demux_.CodecData.data(), 48, demux_.SampleSize, demux_.NumChannels, &cfg_) != ALACGPU_OK)
            throw std::runtime_error("bad codec data");
        int rc = alacgpu_create(&cfg_, 1, device, &ctx_);                                    // new AlacFile + SetInfo (:54-55)
        if (rc != ALACGPU_OK) throw std::runtime_error(std::string("alacgpu_create: ") + alacgpu_strerror(rc));
    }
    ~AlacContext() { if (ctx_) alacgpu_destroy(ctx_); }
    AlacContext(const AlacContext&) = delete;
    AlacContext& operator=(const AlacContext&) = delete;

    int LastSampleNumber = 0;
    int GetSampleRate() const { return demux_.SampleRate != 0 ? demux_.SampleRate : 44100; }
    int GetNumChannels() const { return demux_.NumChannels != 0 ? demux_.NumChannels : 2; }
    int GetBitsPerSample() const { return demux_.SampleSize != 0 ? demux_.SampleSize : 16; }
    int GetBytesPerSample() const { return demux_.SampleSize != 0 ? (demux_.SampleSize + 7) / 8 : 2; }
    const DemuxResT& Demux() const { return demux_; }

    int GetNumSamples() const {   // :108-122
        long long total = 0;
        for (size_t i = 0; i < demux_.SampleByteSize.size(); i++) {
            int size, dur;
            if (!SampleInfo((int)i, size, dur)) return -1;
            total += dur;
        }
        return (int)total;
    }

    // Batch entry point: decodes up to maxPackets packets from the current position; returns the packet count.
    // pcm: int32 per sample, packet p at p*slotInts.
    int ReadBatch(int maxPackets, std::vector<int32_t>& pcm, uint32_t& slotInts, std::vector<int32_t>& outBytes,
                  std::vector<int32_t>& outSamples, std::vector<int32_t>& status, std::vector<int>& durations) {
        std::vector<uint32_t> sizes;
        durations.clear();
        while ((int)sizes.size() < maxPackets) {
            int size, dur;
            if (!SampleInfo(currentSampleBlock_ + (int)sizes.size(), size, dur)) break;
            sizes.push_back((uint32_t)size);
            durations.push_back(dur);
        }
        if (sizes.empty()) return 0;
        std::vector<uint64_t> offsets(sizes.size());
        uint64_t total = 0;
        for (size_t i = 0; i < sizes.size(); i++) { offsets[i] = total; total += sizes[i]; }
        std::vector<uint8_t> blob((size_t)total + 16);
        stream_.Read((size_t)total, blob.data());                       // packets are read in file order (:195)
        slotInts = 16384u * (uint32_t)cfg_.num_channels;
        const size_t n = sizes.size();
        pcm.assign(n * slotInts, 0);
        outBytes.assign(n, 0); outSamples.assign(n, 0); status.assign(n, 0);
        int rc = alacgpu_decode_batch(ctx_, blob.data(), total, offsets.data(), sizes.data(), nullptr, (uint32_t)n, pcm.data(),
                                      slotInts, outBytes.data(), outSamples.data(), status.data());
        if (rc != ALACGPU_OK) throw std::runtime_error(std::string("alacgpu_decode_batch: ") + alacgpu_strerror(rc));
        firstByte_.assign(n, 0);
        for (size_t i = 0; i < n; i++) firstByte_[i] = sizes[i] ? blob[(size_t)offsets[i]] : 0;
        currentSampleBlock_ += (int)n;
        return (int)n;
    }

    // int Read(byte[] buffer): one packet per call, little-endian PCM bytes, 0 at end of stream (:163-172)
    int Read(uint8_t* buffer) {
        if (ready_.empty()) {
            std::vector<int32_t> pcm, ob, os, st;
            std::vector<int> durs;
            uint32_t slot = 0;
            int n = ReadBatch(batchPackets_, pcm, slot, ob, os, st, durs);
            if (n == 0) return 0;
            for (int p = 0; p < n; p++) {
                Packet pk;
                pk.outBytes = ob[(size_t)p]; pk.samples = os[(size_t)p]; pk.status = st[(size_t)p]; pk.duration = durs[(size_t)p];
                // a one-channel element with a prediction type other than 0: the reference skips the predictor without throwing
                // and hands out its output buffer -- behind any compressed frame the residual buffer (AlacFile.cs:484-496 with
                // :486); the library decodes exactly that: an ordinary packet from here on
                if (pk.status == ALACGPU_ST_UNSUPPORTED_PREDTYPE && firstByte_[(size_t)p] >> 5 == 0) pk.status = ALACGPU_ST_OK;
                // a two-channel element of a sample size other than 16 / 24 and 20 / 32: nothing written, no exception (:701-716)
                if (pk.status == ALACGPU_ST_UNSUPPORTED_SAMPLE_SIZE && firstByte_[(size_t)p] >> 5 == 1 && cfg_.sample_size != 20 &&
                    cfg_.sample_size != 32)
                    pk.status = ALACGPU_ST_UNSUPPORTED_ELEMENT;
                size_t cnt = pk.status == ALACGPU_ST_OK ? (size_t)pk.samples * cfg_.num_channels : 0;
                pk.pcm.assign(pcm.begin() + (size_t)p * slot, pcm.begin() + (size_t)p * slot + cnt);
                ready_.push_back(std::move(pk));
            }
        }
        Packet pk = std::move(ready_.front());
        ready_.pop_front();
        ThrowFor(pk.status);                                             // DecodeFrame throws before :198-199 count the packet
        LastSampleNumber += pk.duration;                                 // :199
        std::vector<int32_t> ref(pk.pcm.size() * (cfg_.sample_size == 24 ? 3 : 1) + 8, 0);
        size_t refInts = pk.pcm.empty() ? 0 : alacgpu_expand_reference_layout(&cfg_, pk.pcm.data(), pk.samples, ref.data());
        const int bps = GetBytesPerSample();
        int outBytes = pk.outBytes - offset_ * bps;                      // :200
        const int32_t* src = ref.data() + (offset_ < (int)refInts ? offset_ : (int)refInts);   // :201
        size_t avail = refInts - (size_t)(src - ref.data());
        offset_ = 0;
        if (outBytes <= 0) return outBytes < 0 ? 0 : outBytes;
        size_t needInts = bps == 2 ? (size_t)outBytes / 2 : (size_t)outBytes;
        std::vector<int32_t> tmp(needInts + 1, 0);
        std::memcpy(tmp.data(), src, sizeof(int32_t) * (avail < needInts ? avail : needInts));
        alacgpu_format_samples(bps, tmp.data(), outBytes, buffer);       // FormatSamples (:168)
        return outBytes;
    }

    void SetPosition(long long position) {   // :262-295
        int currentPosition = 0, currentSample = 0;
        for (size_t i = 0; i < demux_.Stsc.size(); i++) {
            const auto& ci = demux_.Stsc[i];
            int lastChunk = i + 1 < demux_.Stsc.size() ? demux_.Stsc[i + 1].FirstChunk : (int)demux_.Stco.size();
            for (int chunk = ci.FirstChunk; chunk <= lastChunk; chunk++) {
                if (chunk - 1 >= (int)demux_.Stco.size()) throw std::out_of_range("Index was outside the bounds of the array.");
                long long pos = demux_.Stco[(size_t)chunk - 1];
                int count = ci.SamplesPerChunk;
                while (count > 0) {
                    int size, dur;
                    if (!SampleInfo(currentSample, size, dur)) break;
                    currentPosition += dur;
                    if (position < currentPosition) {
                        ready_.clear();   // only now: a position past the end is a no-op (:262-295), Read goes on with the prefetched packets
                        stream_.Seek(pos);
                        currentSampleBlock_ = currentSample;
                        LastSampleNumber = currentPosition;
                        offset_ = (int)(position - (currentPosition - dur)) * GetNumChannels();
                        return;
                    }
                    pos += size;
                    currentSample++;
                    count--;
                }
            }
        }
    }

private:
    struct Packet { std::vector<int32_t> pcm; int outBytes = 0, samples = 0, status = 0, duration = 0; };
    bool SampleInfo(int samplenum, int& size, int& dur) const {   // TryGetSampleInfo (:130-156)
        if (samplenum >= (int)demux_.SampleByteSize.size() || demux_.NumTimeToSamples == 0) return false;
        int acc = 0, idx = 0;
        while (demux_.TimeToSample[(size_t)idx].first + acc <= samplenum) {
            acc += demux_.TimeToSample[(size_t)idx].first;
            idx++;
            if (idx >= demux_.NumTimeToSamples) return false;
        }
        size = demux_.SampleByteSize[(size_t)samplenum];
        dur = demux_.TimeToSample[(size_t)idx].second;
        return true;
    }
    void ThrowFor(int status) const {
        switch (status) {
        case ALACGPU_ST_OK: case ALACGPU_ST_UNSUPPORTED_ELEMENT: return;
        case ALACGPU_ST_UNSUPPORTED_SAMPLE_SIZE: throw std::runtime_error("FIXME: unimplemented sample size " + std::to_string(demux_.SampleSize));
        case ALACGPU_ST_UNSUPPORTED_PREDTYPE: throw std::runtime_error("FIXME: unhandled predicition type");
        case ALACGPU_ST_REF_THROWS: throw std::invalid_argument("Destination array was not long enough.");
        case ALACGPU_ST_BAD_SAMPLE_COUNT: case ALACGPU_ST_OVERRUN: throw std::out_of_range("Index was outside the bounds of the array.");
        default: throw std::runtime_error(alacgpu_status_string(status));
        }
    }
    DemuxResT demux_;
    MyStream stream_;
    alacgpu_cfg cfg_{};
    alacgpu_ctx* ctx_ = nullptr;
    int batchPackets_;
    int currentSampleBlock_ = 0, offset_ = 0;
    std::deque<Packet> ready_;
    std::vector<uint8_t> firstByte_;   // of every packet of the last batch (the element's channels field, AlacFile.cs:435)
};

}}  // namespace ALACdotNET::Decoder
