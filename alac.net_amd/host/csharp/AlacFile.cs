// Drop-in replacement for the reference's `internal class AlacFile` (ALACDecoder/AlacFile.cs): same constructor, SetInfo
// and DecodeFrame signatures, so AlacContext.cs:54-55 and :197 of the reference compile unchanged; adds DecodeBatch, which
// the GPU-backed AlacContext.cs next to this file uses.  NOT compiled in this repository's pipeline (no .NET toolchain in
// the image); member for member it mirrors the executed twins: Python `AlacFile` (alac.net_amd/__init__.py) and C++
// `ALACdotNET::Decoder::AlacFile` (alac.net_amd/host/AlacFile.hpp).
using System;

namespace ALACdotNET.Decoder
{
    internal sealed class AlacFile : IDisposable
    {
        private readonly int _samplesize, _numchannels;
        private AlacGpuCfg _cfg;
        private IntPtr _ctx = IntPtr.Zero;

        public AlacFile(int samplesize, int numchannels)      // AlacFile.cs:16-20
        {
            _samplesize = samplesize;
            _numchannels = numchannels;
        }

        internal IntPtr Context => _ctx;
        internal AlacGpuCfg Config => _cfg;

        public void SetInfo(int[] inputbuffer)                 // AlacFile.cs:63-93
        {
            Check(AlacGpuNative.alacgpu_cfg_from_codec_data(inputbuffer, (uint)inputbuffer.Length, _samplesize, _numchannels, out _cfg));
            if (_ctx != IntPtr.Zero) AlacGpuNative.alacgpu_destroy(_ctx);
            Check(AlacGpuNative.alacgpu_create(new[] { _cfg }, 1, 0, out _ctx));
        }

        /// The reference's signature, unchanged (AlacFile.cs:428): fills outbuffer (24-bit: one int per byte), returns the
        /// byte count.  The reference is not told the packet's length; neither is this overload -- it hands the whole read
        /// buffer over.  Callers that know the length (AlacContext.UnpackSamples does: sampleByteSize) use the overload below.
        public int DecodeFrame(byte[] inbuffer, int[] outbuffer) => DecodeFrame(inbuffer, inbuffer.Length, outbuffer);

        public int DecodeFrame(byte[] inbuffer, int packetBytes, int[] outbuffer)
        {
            Check(AlacGpuNative.alacgpu_decode_frame(_ctx, 0, inbuffer, (uint)packetBytes, outbuffer, (uint)outbuffer.Length, out int outBytes, out int status));
            // a one-channel element with a prediction type other than 0: the reference skips the predictor without a word and
            // hands out its output buffer, which behind any compressed frame is the residual buffer (AlacFile.cs:484-496 with
            // :486): alacgpu_decode_frame has written exactly that into outbuffer (status 3 is a warning here)
            if (status == AlacGpuNative.StUnsupportedPredType && (inbuffer[0] >> 5) == 0) return outBytes;
            // a two-channel element of a sample size other than 16 / 24 (decoded) and 20 / 32 (throw): nothing is written (:701-716)
            if (status == AlacGpuNative.StUnsupportedSampleSize && (inbuffer[0] >> 5) == 1 && _cfg.SampleSize != 20 && _cfg.SampleSize != 32)
                return outBytes;
            ThrowFor(status);
            return outBytes;
        }

        /// Batch submit: packet p = blob[offsets[p] .. +sizes[p]) decodes to pcm[p*slotInts ..] (int per sample).
        public void DecodeBatch(byte[] blob, ulong blobBytes, ulong[] offsets, uint[] sizes, uint nPackets, int[] pcm, uint slotInts,
                                int[] outBytes, int[] outSamples, int[] status)
        {
            Check(AlacGpuNative.alacgpu_decode_batch(_ctx, blob, blobBytes, offsets, sizes, null, nPackets, pcm, slotInts, outBytes, outSamples, status));
        }

        /// The reference's exceptions for a per-packet status (AlacFile.cs:574,:650,:660,:715 and the implicit ones).
        internal void ThrowFor(int status)
        {
            switch (status)
            {
                case AlacGpuNative.StOk:
                case AlacGpuNative.StUnsupportedElement: return;   // reference decodes nothing, still returns outputsize
                case AlacGpuNative.StUnsupportedSampleSize: throw new Exception("FIXME: unimplemented sample size " + _cfg.SampleSize);
                case AlacGpuNative.StUnsupportedPredType: throw new Exception("FIXME: unhandled predicition type");
                case AlacGpuNative.StRefThrows: throw new ArgumentException("Destination array was not long enough.");
                case AlacGpuNative.StBadSampleCount:
                case AlacGpuNative.StOverrun: throw new IndexOutOfRangeException();
                default: throw new Exception("unsupported parameter combination");
            }
        }

        internal static void Check(int rc)
        {
            if (rc != 0) throw new InvalidOperationException("alacgpu: " + AlacGpuNative.Error(rc));
        }

        public void Dispose()
        {
            if (_ctx != IntPtr.Zero) { AlacGpuNative.alacgpu_destroy(_ctx); _ctx = IntPtr.Zero; }
        }
    }
}
