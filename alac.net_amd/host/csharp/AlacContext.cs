// GPU-backed `AlacContext` for teekay/ALAC.NET: the reference's public surface (ALACDecoder/AlacContext.cs:20-338) member
// for member -- both constructors, LastSampleNumber, GetSampleRate / GetNumChannels / GetBitsPerSample / GetBytesPerSample /
// GetNumSamples, Read(byte[]), SetPosition(long), Dispose() -- plus the batch entry point north_star asks for
// (ReadBatch, BatchPackets).  It replaces ALACDecoder/AlacContext.cs; QTMovieT.cs, MyStream.cs, DemuxResT.cs, SampleInfo.cs
// and ChunkInfo.cs of the reference stay as they are, and AlacNetNAudioAdapter/ALACFileReader.cs compiles against this
// class unchanged (it only uses the public members above).
//
// What changed behind the surface: UnpackSamples' "read one packet, DecodeFrame it" (AlacContext.cs:194-197) became
// "pre-read the next K packets with the sizes the demuxer holds, ONE alacgpu_decode_batch".  The kernel stores the very
// bytes Read hands out (FormatSamples, AlacContext.cs:214-256, fused into its store: ALACGPU_OUT_PACKED_LE), so Read is a
// copy out of the decoded batch.  Semantics kept on purpose: one packet per Read call, 0 at the end of the stream, the
// post-seek offset arithmetic (in ints of the reference's buffer -- for 24-bit streams that buffer held one int per BYTE,
// App. B Q18 of SURVEY.md) and LastSampleNumber's double count of the seek frame (AlacContext.cs:199,:283).
//
// NOT compiled in this repository's pipeline (no .NET toolchain in the image).  Member for member it mirrors the two
// executed twins -- Python `AlacContext` (alac.net_amd/container.py) and C++ `ALACdotNET::Decoder::AlacContext`
// (alac.net_amd/host/AlacContext.hpp) -- whose tests (tests/test_container.py) run the reference's playback loop, its
// seek, the seek-past-the-end no-op and the 24-bit seek quirk against synthetic .m4a files.
using System;
using System.IO;

namespace ALACdotNET.Decoder
{
    public class AlacContext : IDisposable
    {
        public AlacContext(Stream baseStream, bool disposeStream) : this(baseStream)
        {
            _disposeStream = disposeStream;
        }

        public AlacContext(Stream baseStream)
        {
            _demuxRes = new DemuxResT();
            _inputStream = new BinaryReader(baseStream);
            _myStream = new MyStream(_inputStream);
            var headerRead = new QtMovieT(_myStream, _demuxRes).ReadHeader();
            if (headerRead == MdatPosStatus.None || headerRead == MdatPosStatus.CannotSeekToMdatPosition)
            {
                ReleaseAll(true);
                throw new IOException("Error while loading the QuickTime movie headers.");
            }
            _alac = new AlacFile(_demuxRes.SampleSize, _demuxRes.NumChannels);
            _alac.SetInfo(_demuxRes.CodecData);
            AlacFile.Check(AlacGpuNative.alacgpu_set_output_format(_alac.Context, AlacGpuNative.OutPackedLe));
            _contexts = new[] { _alac.Context };
        }

        /// <summary>New: spread every ReadBatch over the first `gpus` GPUs of the node (alacgpu_device_count tells how many there
        /// are): one more context per extra device, and alacgpu_decode_batch_sharded cuts each batch into contiguous packet
        /// ranges, one native thread per GPU.  Worth it for batches of several thousand packets per GPU.</summary>
        public void UseGpus(int gpus)
        {
            int n = Math.Max(1, Math.Min(gpus, AlacGpuNative.alacgpu_device_count()));
            var list = new IntPtr[n];
            list[0] = _alac.Context;
            var cfg = new[] { _alac.Config };
            for (int d = 1; d < n; d++)
            {
                AlacFile.Check(AlacGpuNative.alacgpu_create(cfg, 1, d, out list[d]));
                AlacFile.Check(AlacGpuNative.alacgpu_set_output_format(list[d], AlacGpuNative.OutPackedLe));
            }
            for (int d = 1; d < _contexts.Length; d++) AlacGpuNative.alacgpu_destroy(_contexts[d]);
            _contexts = list;
        }

        private readonly DemuxResT _demuxRes;
        private readonly AlacFile _alac;
        private readonly BinaryReader _inputStream;
        private readonly MyStream _myStream;
        private readonly bool _disposeStream;
        private bool _disposed;
        private int _currentSampleBlock;     // next packet to FETCH (the decoded queue may hold earlier ones)
        private int _offset;                 // post-seek offset, in ints of the reference's decode buffer
        private IntPtr[] _contexts;          // [0] = _alac's; more after UseGpus

        // ---- the decoded batch: packed little-endian PCM, slot p at p * SlotBytes ----
        private byte[] _blob = new byte[0];
        private ulong[] _offsets = new ulong[0];
        private uint[] _sizes = new uint[0];
        private int[] _pcm = new int[0];      // viewed as bytes through Buffer.BlockCopy
        private int[] _outBytes = new int[0], _outSamples = new int[0], _status = new int[0], _durations = new int[0];
        private int _batchCount, _batchNext;  // packets in the batch / next one Read hands out
        private uint _slotInts;

        /// <summary>Packets fetched and decoded per GPU call (new).  Bigger is faster (the GPU is not full below ~10 000
        /// packets); 256 keeps the memory of a 16-bit stereo stream's batch at 8 MiB.</summary>
        public int BatchPackets { get; set; } = 256;

        /// <summary>Points to the last sample read - can be used to determine position</summary>
        public int LastSampleNumber { get; private set; }

        public int GetSampleRate() => _demuxRes.SampleRate != 0 ? _demuxRes.SampleRate : 44100;
        public int GetNumChannels() => _demuxRes.NumChannels != 0 ? _demuxRes.NumChannels : 2;
        public int GetBitsPerSample() => _demuxRes.SampleSize != 0 ? _demuxRes.SampleSize : 16;
        public int GetBytesPerSample() => _demuxRes.SampleSize != 0 ? (_demuxRes.SampleSize + 7) / 8 : 2;

        /// <summary>Total number of samples in the file, or -1 if some packet has no duration entry</summary>
        public int GetNumSamples()
        {
            int total = 0;
            for (int i = 0; i < _demuxRes.SampleByteSize.Length; i++)
            {
                if (!TrySampleInfo(i, out _, out int duration)) return -1;
                total += duration;
            }
            return total;
        }

        // byte size and duration of packet `samplenum` from stsz / stts (the reference's TryGetSampleInfo, :130-156)
        private bool TrySampleInfo(int samplenum, out int byteSize, out int duration)
        {
            byteSize = duration = 0;
            if (samplenum >= _demuxRes.SampleByteSize.Length || _demuxRes.NumTimeToSamples == 0) return false;
            int before = 0, entry = 0;
            while (_demuxRes.TimeToSample[entry].SampleCount + before <= samplenum)
            {
                before += _demuxRes.TimeToSample[entry].SampleCount;
                if (++entry >= _demuxRes.NumTimeToSamples) return false;
            }
            byteSize = _demuxRes.SampleByteSize[samplenum];
            duration = _demuxRes.TimeToSample[entry].SampleDuration;
            return true;
        }

        /// <summary>
        /// New: fetches up to maxPackets packets from the current position and decodes them in ONE GPU call.  Returns the
        /// number of packets decoded (0 at the end of the stream).  The packets are then handed out by Read, one per call;
        /// PacketBytes(p) exposes them without the copy.
        /// </summary>
        public int ReadBatch(int maxPackets)
        {
            int count = 0;
            long total = 0;
            if (_sizes.Length < maxPackets)
            {
                _sizes = new uint[maxPackets]; _offsets = new ulong[maxPackets];
                _outBytes = new int[maxPackets]; _outSamples = new int[maxPackets]; _status = new int[maxPackets];
                _durations = new int[maxPackets];
            }
            while (count < maxPackets && TrySampleInfo(_currentSampleBlock + count, out int size, out int duration))
            {
                _offsets[count] = (ulong)total;
                _sizes[count] = (uint)size;
                _durations[count] = duration;
                total += size;
                count++;
            }
            _batchCount = 0;
            _batchNext = 0;
            if (count == 0) return 0;
            if (_blob.Length < total + 16) _blob = new byte[total + total / 4 + 16];
            _myStream.Read((int)total, _blob, 0);                        // the packets lie back to back in file order (:195)
            // a slot takes the longest frame the reference can decode (16384 samples per channel, AlacFile.cs:28) unless
            // the stream declares less (frames that carry their own, larger count report BAD_SAMPLE_COUNT -> retry wide)
            uint channels = (uint)GetNumChannels();
            uint declared = _alac.Config.MaxSamplesPerFrame;
            _slotInts = Math.Min(16384u, Math.Max(declared, 1u)) * channels + 8u;
            for (int attempt = 0; attempt < 2; attempt++)
            {
                long need = (long)count * _slotInts;
                if (_pcm.Length < need) _pcm = new int[need];
                AlacFile.Check(AlacGpuNative.alacgpu_decode_batch_sharded(_contexts, (uint)_contexts.Length, _blob, (ulong)total, _offsets, _sizes,
                                                                          null, (uint)count, _pcm, _slotInts, _outBytes, _outSamples, _status));
                bool tooSmall = false;
                for (int p = 0; p < count; p++) tooSmall |= _status[p] == AlacGpuNative.StBadSampleCount;
                if (!tooSmall || _slotInts >= 16384u * channels) break;
                _slotInts = 16384u * channels;
            }
            _currentSampleBlock += count;
            _batchCount = count;
            return count;
        }

        /// <summary>New: packet p of the last batch as (array, byte offset, byte count) of packed little-endian PCM.</summary>
        public void PacketBytes(int p, out int[] slots, out long byteOffset, out int byteCount)
        {
            slots = _pcm;
            byteOffset = (long)p * _slotInts * 4;
            byteCount = _status[p] == AlacGpuNative.StOk ? _outBytes[p] : 0;
        }

        /// <summary>Reads and decodes a single ALAC frame and returns the decoded wave stream (one packet per call;
        /// returns the number of bytes, 0 at the end of the stream) -- served from the GPU-decoded batch.</summary>
        public int Read(byte[] buffer)
        {
            if (_batchNext >= _batchCount && ReadBatch(BatchPackets) == 0) return 0;
            int p = _batchNext++;
            int status = _status[p];
            // one-channel element with a prediction type other than 0: the reference skips the predictor without throwing and
            // hands out its output buffer -- behind any compressed frame the residual buffer (AlacFile.cs:484-496 with :486);
            // the library decoded exactly that (status 3 as a warning): an ordinary packet from here on
            bool staleMono = status == AlacGpuNative.StUnsupportedPredType && (_blob[(long)_offsets[p]] >> 5) == 0;
            // a two-channel element of a sample size other than 16 / 24 and 20 / 32: nothing written, no exception (AlacFile.cs:701-716)
            bool silentStereo = status == AlacGpuNative.StUnsupportedSampleSize && (_blob[(long)_offsets[p]] >> 5) == 1 &&
                                _alac.Config.SampleSize != 20 && _alac.Config.SampleSize != 32;
            // (the reference's DecodeFrame throws BEFORE :198-199 count the packet: LastSampleNumber does not move for a packet
            // that throws.  The packet itself is consumed -- the reference's stream has read past it as well, :195)
            if (!staleMono && !silentStereo) _alac.ThrowFor(status);
            if (staleMono) status = AlacGpuNative.StOk;
            LastSampleNumber += _durations[p];                            // :199
            int bps = GetBytesPerSample();
            int outputBytes = _outBytes[p] - _offset * bps;               // :200
            // :201 drops _offset INTS of the reference's buffer: 16-bit streams hold a sample per int (2 bytes each),
            // 24-bit streams a BYTE per int
            long skipBytes = bps == 2 ? (long)_offset * 2 : _offset;
            _offset = 0;
            if (outputBytes <= 0) return Math.Max(outputBytes, 0);
            if (status == AlacGpuNative.StOk)
            {
                long avail = Math.Max(0, (long)_outBytes[p] - skipBytes);
                int copy = (int)Math.Min(avail, outputBytes);
                Buffer.BlockCopy(_pcm, checked((int)((long)p * _slotInts * 4 + skipBytes)), buffer, 0, copy);
                if (copy < outputBytes) Array.Clear(buffer, copy, outputBytes - copy);
            }
            else
            {
                Array.Clear(buffer, 0, outputBytes);                      // nothing was decoded (:437,:577)
            }
            return outputBytes;
        }

        /// <summary>Sets position in pcm samples</summary>
        public void SetPosition(long position)
        {
            int reached = 0, packet = 0;
            for (int i = 0; i < _demuxRes.Stsc.Length; i++)
            {
                var run = _demuxRes.Stsc[i];
                int lastChunk = i + 1 < _demuxRes.Stsc.Length ? _demuxRes.Stsc[i + 1].FirstChunk : _demuxRes.Stco.Length;
                for (int chunk = run.FirstChunk; chunk <= lastChunk; chunk++)
                {
                    long filePos = _demuxRes.Stco[chunk - 1];
                    for (int left = run.SamplesPerChunk; left > 0; left--)
                    {
                        if (!TrySampleInfo(packet, out int size, out int duration)) break;
                        reached += duration;
                        if (position < reached)
                        {
                            // only now: a position at or past the end is a no-op in the reference, and Read goes on with
                            // the next packet -- which may be sitting, decoded, in the current batch
                            _batchCount = _batchNext = 0;
                            _inputStream.BaseStream.Seek(filePos, SeekOrigin.Begin);
                            _currentSampleBlock = packet;
                            LastSampleNumber = reached;
                            _offset = (int)(position - (reached - duration)) * GetNumChannels();
                            return;
                        }
                        filePos += size;
                        packet++;
                    }
                }
            }
        }

        protected virtual void Dispose(bool disposing)
        {
            ReleaseAll(disposing);
        }

        private void ReleaseAll(bool disposing)
        {
            if (_disposed) return;
            if (_contexts != null)
                for (int d = 1; d < _contexts.Length; d++) AlacGpuNative.alacgpu_destroy(_contexts[d]);
            _alac?.Dispose();
            if (disposing && _disposeStream) _inputStream?.Dispose();
            _disposed = true;
        }

        public void Dispose()
        {
            Dispose(true);
        }
    }
}
