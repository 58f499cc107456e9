// Drop-in replacement for the reference's `internal class AlacFile` (ALACDecoder/AlacFile.cs): same
// constructor, SetInfo and DecodeFrame signatures, so AlacContext.cs:54-55 and :197 compile unchanged;
// adds DecodeBatch.  NOT compiled in this repository's pipeline (no .NET toolchain in the image).
using System;

namespace ALACdotNET.Decoder
{
    internal sealed class AlacFile : IDisposable
    {
        private readonly int _samplesize, _numchannels;
        private AlacGpuCfg _cfg;
        private IntPtr _ctx = IntPtr.Zero;

        public AlacFile(int samplesize, int numchannels)
        {
            _samplesize = samplesize;
            _numchannels = numchannels;
        }

        public void SetInfo(int[] inputbuffer)
        {
            Check(AlacGpuNative.alacgpu_cfg_from_codec_data(inputbuffer, (uint)inputbuffer.Length, _samplesize, _numchannels, out _cfg));
            if (_ctx != IntPtr.Zero) AlacGpuNative.alacgpu_destroy(_ctx);
            Check(AlacGpuNative.alacgpu_create(new[] { _cfg }, 1, 0, out _ctx));
        }

        /// Same contract as the reference: fills outbuffer (24-bit: one int per byte), returns the byte count.
        /// inbuffer is the reused read buffer; packetBytes (new, optional) is the packet's real length.
        public int DecodeFrame(byte[] inbuffer, int[] outbuffer, int packetBytes = -1)
        {
            uint n = (uint)(packetBytes >= 0 ? packetBytes : inbuffer.Length);
            Check(AlacGpuNative.alacgpu_decode_frame(_ctx, 0, inbuffer, n, outbuffer, (uint)outbuffer.Length, out int outBytes, out int status));
            ThrowFor(status);
            return outBytes;
        }

        /// Batch submit: packet p = blob[offsets[p] .. +sizes[p]) decodes to pcm[p*slotInts ..] (int per sample).
        public void DecodeBatch(byte[] blob, ulong[] offsets, uint[] sizes, int[] pcm, uint slotInts, int[] outBytes, int[] outSamples, int[] status)
        {
            Check(AlacGpuNative.alacgpu_decode_batch(_ctx, blob, (ulong)blob.LongLength, offsets, sizes, null, (uint)sizes.Length,
                                                    pcm, slotInts, outBytes, outSamples, status));
        }

        private void ThrowFor(int status)
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

        private static void Check(int rc)
        {
            if (rc != 0) throw new InvalidOperationException("alacgpu: " + AlacGpuNative.Error(rc));
        }

        public void Dispose()
        {
            if (_ctx != IntPtr.Zero) { AlacGpuNative.alacgpu_destroy(_ctx); _ctx = IntPtr.Zero; }
        }
    }
}
