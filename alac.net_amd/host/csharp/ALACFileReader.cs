// NAudio adapter over the GPU-backed AlacContext.  The reference's own AlacNetNAudioAdapter/ALACFileReader.cs compiles
// UNCHANGED against host/csharp/AlacContext.cs (it only uses AlacContext's public members) -- this file is the same
// public surface (both constructors, WaveFormat, Length, Position, Read(buffer, offset, count), Dispose(bool)) with one
// addition: a constructor argument for the batch size, so that a player that reads a few KiB at a time and a converter
// that reads whole files can both pick how many packets go to the GPU per call.
// NOT compiled in this repository's pipeline (no .NET toolchain; NAudio 1.10.0 is not in the image).  Executed twins:
// alac.net_amd/naudio_adapter.py (Python) and alac.net_amd/host/ALACFileReader.hpp (C++), tests/test_container.py.
using System;
using System.IO;
using ALACdotNET.Decoder;
using NAudio.Wave;

namespace AlacNetNAudioAdapter
{
    public class ALACFileReader : WaveStream
    {
        /// <summary>The underlying stream will NOT be disposed after use</summary>
        public ALACFileReader(Stream baseStream) : this(baseStream, false)
        {
        }

        public ALACFileReader(Stream baseStream, bool disposeAfterUse) : this(baseStream, disposeAfterUse, 256)
        {
        }

        /// <summary>New: batchPackets = packets fetched and decoded per GPU call</summary>
        public ALACFileReader(Stream baseStream, bool disposeAfterUse, int batchPackets)
        {
            _alacContext = new AlacContext(baseStream, disposeAfterUse) { BatchPackets = Math.Max(1, batchPackets) };
            _waveFormat = new WaveFormat(_alacContext.GetSampleRate(), _alacContext.GetBytesPerSample() * 8, _alacContext.GetNumChannels());
            Length = (long)_alacContext.GetNumSamples() * _waveFormat.BlockAlign;
            _frame = new byte[65546 * _waveFormat.BitsPerSample / 8 * _waveFormat.Channels];
        }

        private readonly WaveFormat _waveFormat;
        private readonly AlacContext _alacContext;
        private readonly byte[] _frame;          // the packet Read is handing out, and how much of it is left
        private int _frameLeft, _frameAt;
        private readonly object _gate = new object();

        public override long Length { get; }

        public override long Position
        {
            get => (long)_alacContext.LastSampleNumber * _waveFormat.BlockAlign;
            set
            {
                lock (_gate)
                {
                    _alacContext.SetPosition(value / _waveFormat.BlockAlign);
                    _frameLeft = 0;              // nothing more comes out of the packet that was being handed out
                }
            }
        }

        public override WaveFormat WaveFormat => _waveFormat;

        public override int Read(byte[] buffer, int offset, int count)
        {
            int done = 0;
            lock (_gate)
            {
                while (done < count)
                {
                    if (_frameLeft == 0)
                    {
                        _frameAt = 0;
                        _frameLeft = _alacContext.Read(_frame);
                        if (_frameLeft == 0) break;              // end of the stream
                    }
                    int take = Math.Min(_frameLeft, count - done);
                    Buffer.BlockCopy(_frame, _frameAt, buffer, offset + done, take);
                    _frameAt += take;
                    _frameLeft -= take;
                    done += take;
                }
            }
            return done;
        }

        protected override void Dispose(bool disposing)
        {
            if (!disposing) return;
            lock (_gate)
            {
                _alacContext.Dispose();
            }
        }
    }
}
