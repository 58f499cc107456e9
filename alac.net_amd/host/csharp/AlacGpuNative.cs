// P/Invoke binding of include/alacgpu.h for the C# host (drop into ALACDecoder/).
// NOT compiled or executed in this repository's pipeline: the build image has no .NET toolchain.
// It is kept mechanical -- blittable arguments only -- and mirrors the executed ctypes binding in
// alac.net_amd/__init__.py (SYMBOLS table) one to one.
using System;
using System.Runtime.InteropServices;

namespace ALACdotNET.Decoder
{
    [StructLayout(LayoutKind.Sequential, Pack = 1, Size = 12)]
    internal struct AlacGpuCfg
    {
        public uint MaxSamplesPerFrame;   // CodecData[24..27]
        public byte SampleSize;           // CodecData[29]
        public byte RiceHistoryMult;      // CodecData[30]
        public byte RiceInitialHistory;   // CodecData[31]
        public byte RiceKModifier;        // CodecData[32]
        public byte NumChannels;          // AlacFile ctor arg
        public byte CtorSampleSize;       // AlacFile ctor arg
        public byte Reserved;
    }

    internal static class AlacGpuNative
    {
        private const string Lib = "alacgpu";   // libalacgpu.so

        public const int StOk = 0, StUnsupportedElement = 1, StUnsupportedSampleSize = 2, StUnsupportedPredType = 3,
                         StBadSampleCount = 4, StOverrun = 5, StRefThrows = 6, StUnsupportedParams = 7;
        public const int OutInt32 = 0, OutPackedLe = 1;

        [DllImport(Lib)] public static extern int alacgpu_version();
        [DllImport(Lib)] public static extern int alacgpu_device_count();
        [DllImport(Lib)] public static extern int alacgpu_create([In] AlacGpuCfg[] cfgs, uint nCfgs, int device, out IntPtr ctx);
        [DllImport(Lib)] public static extern void alacgpu_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern int alacgpu_cfg_from_codec_data([In] int[] codecData, uint nInts, int samplesize, int numchannels, out AlacGpuCfg cfg);
        [DllImport(Lib)] public static extern int alacgpu_decode_batch(IntPtr ctx, [In] byte[] blob, ulong blobBytes,
            [In] ulong[] offsets, [In] uint[] sizes, [In] ushort[] cfgIdx, uint nPackets,
            [Out] int[] pcmOut, uint slotInts, [Out] int[] outBytes, [Out] int[] outSamples, [Out] int[] status);
        /// <summary>One batch over several contexts (one per GPU, alacgpu_device_count) from this process: contiguous packet
        /// ranges, one native thread per context, every range writes its own part of the arrays.</summary>
        [DllImport(Lib)] public static extern int alacgpu_decode_batch_sharded([In] IntPtr[] ctxs, uint nCtxs, [In] byte[] blob, ulong blobBytes,
            [In] ulong[] offsets, [In] uint[] sizes, [In] ushort[] cfgIdx, uint nPackets,
            [Out] int[] pcmOut, uint slotInts, [Out] int[] outBytes, [Out] int[] outSamples, [Out] int[] status);
        // the same entry point over raw pointers (pinned / alacgpu_alloc_pinned memory; packed output viewed as bytes)
        [DllImport(Lib, EntryPoint = "alacgpu_decode_batch")] public static extern int alacgpu_decode_batch_ptr(IntPtr ctx, IntPtr blob, ulong blobBytes,
            [In] ulong[] offsets, [In] uint[] sizes, [In] ushort[] cfgIdx, uint nPackets,
            IntPtr pcmOut, uint slotInts, [Out] int[] outBytes, [Out] int[] outSamples, [Out] int[] status);
        [DllImport(Lib)] public static extern int alacgpu_decode_frame(IntPtr ctx, uint cfgIndex, [In] byte[] inbuffer, uint inBytes,
            [Out] int[] outbuffer, uint outCapacityInts, out int outBytes, out int status);
        /// <summary>0: one int per sample (default); 1: packed little-endian PCM, the bytes AlacContext.Read returns
        /// (AlacContext.FormatSamples fused into the store) at the start of every slot; out_bytes[p] of them.</summary>
        [DllImport(Lib)] public static extern int alacgpu_set_output_format(IntPtr ctx, int format);
        [DllImport(Lib)] public static extern IntPtr alacgpu_alloc_pinned(UIntPtr bytes);
        [DllImport(Lib)] public static extern void alacgpu_free_pinned(IntPtr p);
        [DllImport(Lib)] public static extern float alacgpu_last_kernel_ms(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr alacgpu_status_string(int status);
        [DllImport(Lib)] public static extern IntPtr alacgpu_strerror(int rc);
        [DllImport(Lib)] public static extern IntPtr alacgpu_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern int alacgpu_ctx_device(IntPtr ctx);

        // ---- multi-GPU, one process per GPU: the packet partition and the RCCL all-gather of decoded PCM (include/alacgpu.h) ----
        public const int CommIdBytes = 128;
        /// <summary>first[world + 1]: rank r owns packets first[r] .. first[r+1] (whole groups of 8, balanced by packet bytes).</summary>
        [DllImport(Lib)] public static extern int alacgpu_shard_ranges([In] uint[] sizes, uint nPackets, uint world, [Out] uint[] first);
        /// <summary>Rank 0: 128 bytes to hand to the other ranks (pipe, socket, file ...) before alacgpu_comm_create.</summary>
        [DllImport(Lib)] public static extern int alacgpu_comm_get_unique_id([Out] byte[] id128);
        /// <summary>Collective (ncclCommInitRank): every rank calls it with the same id.</summary>
        [DllImport(Lib)] public static extern int alacgpu_comm_create(IntPtr ctx, [In] byte[] id128, int rank, int world, out IntPtr comm);
        [DllImport(Lib)] public static extern void alacgpu_comm_destroy(IntPtr comm);
        [DllImport(Lib)] public static extern int alacgpu_comm_rank(IntPtr comm);
        [DllImport(Lib)] public static extern int alacgpu_comm_world(IntPtr comm);
        [DllImport(Lib)] public static extern IntPtr alacgpu_comm_last_error(IntPtr comm);
        /// <summary>dFullPcm (device memory, the whole batch's slots in global packet order) holds this rank's packets decoded
        /// in place; on return -- asynchronous on hipStream -- every rank holds every packet (ncclAllGather, int32).</summary>
        [DllImport(Lib)] public static extern int alacgpu_allgather_pcm(IntPtr comm, IntPtr dFullPcm, [In] uint[] first, uint slotInts, IntPtr hipStream);
        /// <summary>Decode this rank's range in up to four pieces and gather piece k while piece k + 1 decodes; all device arrays are
        /// indexed by GLOBAL packet number.</summary>
        [DllImport(Lib)] public static extern int alacgpu_decode_allgather_device(IntPtr ctx, IntPtr comm, IntPtr dBlob, ulong blobBytes,
            IntPtr dOffsets, IntPtr dSizes, IntPtr dCfgIdx, [In] uint[] first, IntPtr dFullPcm, uint slotInts,
            IntPtr dOutBytes, IntPtr dOutSamples, IntPtr dStatus, uint nChunks, IntPtr hipStream);

        public static string Error(int rc) => Marshal.PtrToStringAnsi(alacgpu_strerror(rc));
    }
}
