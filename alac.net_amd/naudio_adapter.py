"""Executed mirror of the reference's NAudio adapter, `AlacNetNAudioAdapter.ALACFileReader`
(AlacNetNAudioAdapter/ALACFileReader.cs:22-126), over the GPU-backed `AlacContext` of container.py.

NAudio itself (`WaveStream`, `WaveFormat`) is a Windows audio library that is neither in the image nor on the path;
`WaveFormat` below carries just the fields the adapter and its callers touch (ALACFileReader.cs:42-44, Program.cs:41).
The C# adapter of the reference compiles UNCHANGED against the C# `AlacContext` in host/csharp/ (same public members);
this Python class and the C++ twin (host/ALACFileReader.hpp) are what the tests execute.
"""
import threading

from .container import AlacContext


class WaveFormat:
    """PCM WaveFormat(rate, bits, channels) as NAudio computes it."""

    def __init__(self, rate, bits, channels):
        self.SampleRate, self.BitsPerSample, self.Channels = int(rate), int(bits), int(channels)
        self.BlockAlign = self.Channels * (self.BitsPerSample // 8)
        self.AverageBytesPerSecond = self.SampleRate * self.BlockAlign

    def __repr__(self):
        return f"{self.BitsPerSample} bit PCM: {self.SampleRate // 1000}kHz {self.Channels} channels"


class ALACFileReader:
    """WaveStream-shaped reader: WaveFormat, Length, Position (get/set), Read(buffer, offset, count), Dispose."""

    def __init__(self, baseStream, disposeAfterUse=False, device=0, batch_packets=256):
        self._alacContext = AlacContext(baseStream, disposeAfterUse, device=device, batch_packets=batch_packets)   # :41
        c = self._alacContext
        self._waveFormat = WaveFormat(c.GetSampleRate(), c.GetBytesPerSample() * 8, c.GetNumChannels())            # :42
        self.Length = c.GetNumSamples() * self._waveFormat.BlockAlign                                              # :43
        self._decompressBuffer = bytearray(65546 * self._waveFormat.BitsPerSample // 8 * self._waveFormat.Channels)  # :44
        self._decompressLeftovers = 0
        self._decompressBufferOffset = 0
        self._repositionLock = threading.Lock()                                                                   # :53

    @property
    def WaveFormat(self):
        return self._waveFormat

    @property
    def Position(self):                                            # :65
        return self._alacContext.LastSampleNumber * self._waveFormat.BlockAlign

    @Position.setter
    def Position(self, value):                                     # :66-73
        with self._repositionLock:
            self._alacContext.SetPosition(value // self._waveFormat.BlockAlign)
            self._decompressLeftovers = 0   # after repositioning no more data comes from the buffer

    def Read(self, buffer, offset, count):                         # :89-116
        bytesRead = 0
        with self._repositionLock:
            while bytesRead < count:
                if self._decompressLeftovers > 0:
                    toCopy = min(self._decompressLeftovers, count - bytesRead)
                    o = self._decompressBufferOffset
                    buffer[offset:offset + toCopy] = self._decompressBuffer[o:o + toCopy]
                    self._decompressLeftovers -= toCopy
                    self._decompressBufferOffset = 0 if self._decompressLeftovers == 0 else o + toCopy
                    bytesRead += toCopy
                    offset += toCopy
                if bytesRead >= count:
                    break
                self._decompressBufferOffset = 0
                bytesUnpacked = self._alacContext.Read(self._decompressBuffer)
                if bytesUnpacked == 0:
                    break
                self._decompressLeftovers += bytesUnpacked
        return bytesRead

    def Dispose(self):                                             # :118-125
        with self._repositionLock:
            self._alacContext.Dispose()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.Dispose()
