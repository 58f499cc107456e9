// AlacFile.hpp -- C++ host-side mirror of the reference's decoder surface for the frame-decode path,
// on top of the C ABI (include/alacgpu.h).  The reference is C# (ALACDecoder/AlacFile.cs); no .NET
// toolchain exists in this image, so the executed host mirror is C++ (and Python, alac.net_amd/__init__.py);
// the C# binding a maintainer would add is in host/csharp/ and INTEGRATION.md.
//
// Same names, argument meaning and error behaviour as the reference:
//   AlacFile(int samplesize, int numchannels)              AlacFile.cs:16
//   void SetInfo(const int* inputbuffer)                   AlacFile.cs:63   (int-per-byte CodecData, >= 48 ints)
//   int  DecodeFrame(const uint8_t* inbuffer, size_t, int* outbuffer)   AlacFile.cs:428 (byte count returned)
// plus the batch-submit entry point north_star asks for (DecodeBatch).  Exceptions carry the reference's
// messages ("FIXME: unimplemented sample size N", "FIXME: unhandled predicition type: ...").
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "alacgpu.h"

namespace ALACdotNET { namespace Decoder {

class AlacFile {
public:
    AlacFile(int samplesize, int numchannels, int device = 0)
        : samplesize_(samplesize), numchannels_(numchannels), device_(device) {}
    ~AlacFile() { if (ctx_) alacgpu_destroy(ctx_); }
    AlacFile(const AlacFile&) = delete;
    AlacFile& operator=(const AlacFile&) = delete;

    void SetInfo(const int32_t* inputbuffer, uint32_t n_ints = 48) {
        if (alacgpu_cfg_from_codec_data(inputbuffer, n_ints, samplesize_, numchannels_, &cfg_) != ALACGPU_OK)
            throw std::invalid_argument("SetInfo: codec data too short");
        if (ctx_) { alacgpu_destroy(ctx_); ctx_ = nullptr; }
        int rc = alacgpu_create(&cfg_, 1, device_, &ctx_);
        if (rc != ALACGPU_OK) throw std::runtime_error(std::string("alacgpu_create: ") + alacgpu_strerror(rc));
    }

    // outbuffer receives the reference's own int[] layout (24-bit: one int per byte); capacity in ints.
    int DecodeFrame(const uint8_t* inbuffer, uint32_t in_bytes, int32_t* outbuffer, uint32_t out_capacity = 1024 * 80) {
        require_ctx();
        int32_t out_bytes = 0, status = 0;
        int rc = alacgpu_decode_frame(ctx_, 0, inbuffer, in_bytes, outbuffer, out_capacity, &out_bytes, &status);
        if (rc != ALACGPU_OK) throw std::runtime_error(std::string("alacgpu_decode_frame: ") + alacgpu_strerror(rc));
        // a one-channel element with a prediction type other than 0: the reference skips the predictor without a word and hands
        // out its output buffer, which behind any compressed frame is the residual buffer (AlacFile.cs:484-496 with :486): the
        // library has written exactly that (the un-predicted residuals) into outbuffer, with status 3 as a warning
        if (status == ALACGPU_ST_UNSUPPORTED_PREDTYPE && in_bytes > 0 && (inbuffer[0] >> 5) == 0) return out_bytes;
        // a two-channel element of a sample size other than 16 / 24 (decoded) and 20 / 32 (throw): nothing written (:701-716)
        if (status == ALACGPU_ST_UNSUPPORTED_SAMPLE_SIZE && in_bytes > 0 && (inbuffer[0] >> 5) == 1 && cfg_.sample_size != 20 &&
            cfg_.sample_size != 32)
            return out_bytes;
        throw_for(status);
        return out_bytes;  // AlacFile.cs:718
    }

    // Batch submit: packet p = blob[offsets[p], +sizes[p]) -> pcm_out + p*slot_ints (int32 per sample).
    // Per-packet problems are reported in status[] instead of exceptions.
    void DecodeBatch(const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offsets, const uint32_t* sizes,
                     uint32_t n_packets, int32_t* pcm_out, uint32_t slot_ints, int32_t* out_bytes, int32_t* out_samples,
                     int32_t* status) {
        require_ctx();
        int rc = alacgpu_decode_batch(ctx_, blob, blob_bytes, offsets, sizes, nullptr, n_packets, pcm_out, slot_ints,
                                      out_bytes, out_samples, status);
        if (rc != ALACGPU_OK)
            throw std::runtime_error(std::string("alacgpu_decode_batch: ") + alacgpu_strerror(rc) + " " + alacgpu_last_error(ctx_));
    }

    const alacgpu_cfg& Config() const { return cfg_; }

private:
    void require_ctx() const { if (!ctx_) throw std::logic_error("SetInfo must be called first"); }
    void throw_for(int status) const {
        switch (status) {
        case ALACGPU_ST_OK:
        case ALACGPU_ST_UNSUPPORTED_ELEMENT:  // reference decodes nothing and still returns outputsize (:437,:577,:718)
            return;
        case ALACGPU_ST_UNSUPPORTED_SAMPLE_SIZE:
            throw std::runtime_error("FIXME: unimplemented sample size " + std::to_string((int)cfg_.sample_size));  // :574,:715
        case ALACGPU_ST_UNSUPPORTED_PREDTYPE:
            throw std::runtime_error("FIXME: unhandled predicition type");                                          // :650,:660
        case ALACGPU_ST_REF_THROWS:
            throw std::invalid_argument("Destination array was not long enough.");                                  // :265
        case ALACGPU_ST_BAD_SAMPLE_COUNT:
        case ALACGPU_ST_OVERRUN:
            throw std::out_of_range("Index was outside the bounds of the array.");                                  // :242
        default:
            throw std::runtime_error(alacgpu_status_string(status));
        }
    }
    int samplesize_, numchannels_, device_;
    alacgpu_cfg cfg_{};
    alacgpu_ctx* ctx_ = nullptr;
};

}}  // namespace ALACdotNET::Decoder
