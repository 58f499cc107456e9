"""SECOND CPU ORACLE, deliberately literal -- TEST INFRASTRUCTURE ONLY (same rules as alac_oracle.c: only tests/ and the
golden-vector generator may import it; the product never does).

A statement-by-statement Python restatement of /root/reference/ALACDecoder/AlacFile.cs (teekay/ALAC.NET), written as a
second, independent reading of the reference next to the optimised C restatement in alac_oracle.c: it keeps the reference's
own shape -- one stateful decoder object with reused scratch arrays, the bit reader's three-byte loads, the unary prefix
read one bit per call through a recursive local function, CountLeadingZeros as a list of (condition, action) closures
searched in order, the predictor's per-sample delegates -- instead of the closed forms the C oracle and the kernels use.
tests/golden/make_golden.py decodes the fixture packets with THIS file; tests then hold the C oracle and the HIP path to
those outputs, so a misreading shared by the C oracle, the encoder and the kernels does not cancel out silently.
"Parity unpinned" still applies (the reference ships no fixtures and cannot run here): this narrows the risk, it does not
remove it.  Every function cites the reference lines it follows.

C# semantics reproduced explicitly: `int` wraps at 32 bits, `>>` on int is arithmetic, shift counts are taken modulo 32,
`/` truncates toward zero, array indexing outside the bounds throws (IndexError here), Array.Copy with a length that does
not fit throws ArgumentException (ValueError here).
"""


def i32(x):
    """wrap to a C# int"""
    x &= 0xFFFFFFFF
    return x - 0x100000000 if x & 0x80000000 else x


def shl(x, n):
    """C# `x << n` on int"""
    return i32((x & 0xFFFFFFFF) << (n & 31))


def sar(x, n):
    """C# `x >> n` on int (arithmetic; Python's >> on a negative int is arithmetic as well)"""
    return i32(x) >> (n & 31)


def shr_u(x, n):
    """C# `x >> n` on uint"""
    return (x & 0xFFFFFFFF) >> (n & 31)


def cs_div(a, b):
    """C# integer division: truncates toward zero"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


class CsArray(list):
    """a C# array: fixed length, IndexOutOfRangeException outside [0, Length)"""

    def __getitem__(self, i):
        if not 0 <= i < len(self):
            raise IndexError("Index was outside the bounds of the array.")
        return list.__getitem__(self, i)

    def __setitem__(self, i, v):
        if not 0 <= i < len(self):
            raise IndexError("Index was outside the bounds of the array.")
        list.__setitem__(self, i, v)


def new_int_array(n):
    return CsArray([0] * n)


def array_copy(src, src_index, dst, dst_index, length):
    """System.Array.Copy"""
    if length < 0 or src_index + length > len(src) or dst_index + length > len(dst):
        raise ValueError("Destination array was not long enough. Check destIndex and length, and the array's lower bounds.")
    tmp = [list.__getitem__(src, src_index + k) for k in range(length)]
    for k in range(length):
        list.__setitem__(dst, dst_index + k, tmp[k])


class AlacFile:
    BufferSize = 16384       # AlacFile.cs:28
    RiceThreshold = 8        # :61

    def __init__(self, samplesize, numchannels):                         # :16-20
        self._numchannels = numchannels
        self._bytespersample = cs_div(samplesize, 8) * numchannels
        self._inputBuffer = None
        self._ibIdx = 0
        self._inputBufferBitaccumulator = 0
        B = self.BufferSize
        self._predicterrorBufferA = new_int_array(B)                     # :30-36
        self._predicterrorBufferB = new_int_array(B)
        self._outputsamplesBufferA = new_int_array(B)
        self._outputsamplesBufferB = new_int_array(B)
        self._uncompressedBytesBufferA = new_int_array(B)
        self._uncompressedBytesBufferB = new_int_array(B)
        self._setinfoMaxSamplesPerFrame = 0
        self._setinfoSampleSize = 0
        self._setinfoRiceHistorymult = 0
        self._setinfoRiceInitialhistory = 0
        self._setinfoRiceKmodifier = 0
        self._predictorCoefTable = new_int_array(1024)                   # :58-60
        self._predictorCoefTableA = new_int_array(1024)
        self._predictorCoefTableB = new_int_array(1024)

    # ---- :63-93 --------------------------------------------------------------------------------------------------
    def SetInfo(self, inputbuffer):
        ptrIndex = 0
        ptrIndex += 4  # size
        ptrIndex += 4  # frma
        ptrIndex += 4  # alac
        ptrIndex += 4  # size
        ptrIndex += 4  # alac
        ptrIndex += 4  # 0 ?
        self._setinfoMaxSamplesPerFrame = i32(shl(inputbuffer[ptrIndex], 24) + shl(inputbuffer[ptrIndex + 1], 16) +
                                              shl(inputbuffer[ptrIndex + 2], 8) + inputbuffer[ptrIndex + 3])
        ptrIndex += 4
        ptrIndex += 1  # _setinfo_7A
        self._setinfoSampleSize = inputbuffer[ptrIndex]
        ptrIndex += 1
        self._setinfoRiceHistorymult = inputbuffer[ptrIndex] & 0xff
        ptrIndex += 1
        self._setinfoRiceInitialhistory = inputbuffer[ptrIndex] & 0xff
        ptrIndex += 1
        self._setinfoRiceKmodifier = inputbuffer[ptrIndex] & 0xff
        # (the remaining fields are stored and never read: :83-92)

    # ---- stream reading, :101-152 ----------------------------------------------------------------------------------
    def _byte(self, idx):
        # byte[] indexing: IndexOutOfRangeException past the 80 KiB read buffer (AlacContext.cs:64)
        if not 0 <= idx < len(self._inputBuffer):
            raise IndexError("Index was outside the bounds of the array.")
        return self._inputBuffer[idx]

    def Readbits16(self, bits):                                          # :101-118
        part1 = self._byte(self._ibIdx) & 0xff
        part2 = self._byte(self._ibIdx + 1) & 0xff
        part3 = self._byte(self._ibIdx + 2) & 0xff
        result = sar(shl((part1 << 16) | (part2 << 8) | part3, self._inputBufferBitaccumulator) & 0x00ffffff, 24 - bits)
        newAccumulator = self._inputBufferBitaccumulator + bits
        self._ibIdx += newAccumulator >> 3
        self._inputBufferBitaccumulator = newAccumulator & 7
        return result

    def Readbits(self, bitsParam):                                       # :125-129
        bits = bitsParam if bitsParam <= 16 else bitsParam - 16
        # left operand first (C# evaluates left to right): the high half is read before the low half
        high = 0 if bitsParam <= 16 else shl(self.Readbits16(16), bits)
        return i32(high | self.Readbits16(bits))

    def Readbit(self):                                                   # :135-143
        part1 = self._byte(self._ibIdx) & 0xff
        result = (shl(part1, self._inputBufferBitaccumulator) >> 7) & 1
        newAccumulator = self._inputBufferBitaccumulator + 1
        self._ibIdx += cs_div(newAccumulator, 8)
        self._inputBufferBitaccumulator = newAccumulator % 8
        return result

    def Unreadbits(self, bits):                                          # :145-152
        newAccumulator = self._inputBufferBitaccumulator - bits
        self._ibIdx += newAccumulator >> 3
        self._inputBufferBitaccumulator = newAccumulator & 7
        if self._inputBufferBitaccumulator < 0:
            self._inputBufferBitaccumulator *= -1

    # ---- :154-191 ------------------------------------------------------------------------------------------------------
    def CountLeadingZerosExtra(self, curbyteParam, initialZeroes):
        initialCondition = (curbyteParam & 0xf0) == 0
        zeroes = initialZeroes + 4 if initialCondition else initialZeroes
        curbyte = sar(curbyteParam, 4) if not initialCondition else curbyteParam
        conditions = [
            (lambda: (curbyte & 0x8) != 0, 0),
            (lambda: (curbyte & 0x4) != 0, 1),
            (lambda: (curbyte & 0x2) != 0, 2),
            (lambda: (curbyte & 0x1) != 0, 3),
            (lambda: True, 4),        # "shouldn't get here"
        ]
        for condition, output in conditions:                             # conditions.First(c => c.Condition())
            if condition():
                return output + zeroes
        raise AssertionError("unreachable")

    def CountLeadingZeros(self, input_):
        output = 0
        curbyte = sar(input_, 24)
        conditions = [
            (lambda x: x != 0, lambda: sar(input_, 16)),
            (lambda x: (x & 0xFF) != 0, lambda: sar(input_, 8)),
            (lambda x: (x & 0xFF) != 0, lambda: input_),
            (lambda x: (x & 0xFF) != 0, lambda: 0),
        ]
        for returnIf, setCurbyteTo in conditions:
            if returnIf(curbyte):
                return self.CountLeadingZerosExtra(curbyte, output)
            output += 8
            curbyte = setCurbyteTo()
        return output + 8

    # ---- :193-212, :254 ---------------------------------------------------------------------------------------------------
    def EntropyDecodeValue(self, readSampleSize, k, riceKmodifierMask):
        riceKmodifierMask = i32(riceKmodifierMask)                       # the uint overload casts to int (:254)

        def incrementUntil(x):
            return incrementUntil(x + 1) if x <= self.RiceThreshold and self.Readbit() != 0 else x

        decodedValue = incrementUntil(0)
        if decodedValue > self.RiceThreshold:
            return self.Readbits(readSampleSize) & i32(shr_u(0xffffffff, 32 - readSampleSize))
        if k == 1:
            return decodedValue
        extraBits = self.Readbits(k)
        decodedValue = i32(decodedValue * (i32(shl(1, k) - 1) & riceKmodifierMask))
        if extraBits > 1:
            decodedValue = i32(decodedValue + extraBits - 1)
        else:
            self.Unreadbits(1)
        return decodedValue

    # ---- :214-252 -------------------------------------------------------------------------------------------------------
    def EntropyRiceDecode(self, outputBuffer, outputSize, readSampleSize, riceInitialhistory, riceKmodifier, riceHistorymult,
                          riceKmodifierMask):
        history = riceInitialhistory
        outputCount = 0
        signModifier = 0
        while outputCount < outputSize:
            initialK = 31 - riceKmodifier - self.CountLeadingZeros(i32(sar(history, 9) + 3))
            k = initialK + riceKmodifier if initialK < 0 else riceKmodifier
            decodedValue = i32(self.EntropyDecodeValue(readSampleSize, k, 0xFFFFFFFF) + signModifier)
            almostFinalValue = cs_div(i32(decodedValue + 1), 2)
            outputBuffer[outputCount] = i32(almostFinalValue * -1) if (decodedValue & 1) != 0 else almostFinalValue
            signModifier = 0
            history = 0xFFFF if decodedValue > 0xFFFF else \
                i32(i32(history + i32(decodedValue * riceHistorymult)) - sar(i32(history * riceHistorymult), 9))
            if history < 128 and outputCount + 1 < outputSize:
                signModifier = 1
                k = self.CountLeadingZeros(history) + cs_div(history + 16, 64) - 24
                blockSize = self.EntropyDecodeValue(16, k, riceKmodifierMask)
                if blockSize > 0:
                    for j in range(blockSize):
                        outputBuffer[outputCount + 1 + j] = 0
                    outputCount += blockSize
                if blockSize > 0xFFFF:
                    signModifier = 0
                history = 0
            outputCount += 1

    # ---- :256-336 -------------------------------------------------------------------------------------------------------
    def PredictorDecompressFirAdapt(self, errorBuffer, outputSize, readsamplesize, predictorCoefTable, predictorCoefNum,
                                    predictorQuantitization):
        bufferOut = errorBuffer                                          # :260 same array
        if predictorCoefNum == 0:
            if outputSize <= 1:
                return bufferOut
            sizeToCopy = (outputSize - 1) * 4
            array_copy(errorBuffer, 1, bufferOut, 1, sizeToCopy)         # :265 (counts ELEMENTS)
            return bufferOut
        if predictorCoefNum == 0x1f:
            if outputSize <= 1:
                return bufferOut
            for i in range(outputSize - 1):
                prevValue = bufferOut[i]
                errorValue = errorBuffer[i + 1]
                bitsmove = 32 - readsamplesize
                bufferOut[i + 1] = sar(shl(i32(prevValue + errorValue), bitsmove), bitsmove)
            return bufferOut
        if predictorCoefNum > 0:                                         # warm-up :284-293
            for i in range(predictorCoefNum):
                val = i32(bufferOut[i] + errorBuffer[i + 1])
                bitsmove = 32 - readsamplesize
                val = sar(shl(val, bitsmove), bitsmove)
                bufferOut[i + 1] = val
        if predictorCoefNum <= 0:
            return bufferOut
        bufferOutIdx = 0
        for i in range(predictorCoefNum + 1, outputSize):
            sum_ = 0
            errorVal = errorBuffer[i]
            for j in range(predictorCoefNum):
                sum_ = i32(sum_ + i32(i32(bufferOut[bufferOutIdx + predictorCoefNum - j] - bufferOut[bufferOutIdx]) *
                                      predictorCoefTable[j]))
            outval = i32(shl(1, predictorQuantitization - 1) + sum_)
            outval = sar(outval, predictorQuantitization)
            outval = i32(i32(outval + bufferOut[bufferOutIdx]) + errorVal)
            bitsmove = 32 - readsamplesize
            outval = sar(shl(outval, bitsmove), bitsmove)
            bufferOut[bufferOutIdx + predictorCoefNum + 1] = outval
            if errorVal != 0:
                whileValGt0 = lambda v: v > 0                             # noqa: E731  (:314-319 delegates)
                whileValLt0 = lambda v: v < 0                             # noqa: E731
                conditionToUse = whileValGt0 if errorVal > 0 else whileValLt0
                intAsIs = lambda x: x                                     # noqa: E731
                intNeg = lambda x: i32(-x)                                # noqa: E731
                intPosOrNeg = intAsIs if conditionToUse is whileValGt0 else intNeg
                predictorNum = predictorCoefNum - 1
                while predictorNum >= 0 and conditionToUse(errorVal):
                    val = i32(bufferOut[bufferOutIdx] - bufferOut[bufferOutIdx + predictorCoefNum - predictorNum])
                    sign = intPosOrNeg(-1 if val < 0 else (1 if val > 0 else 0))
                    predictorCoefTable[predictorNum] = i32(predictorCoefTable[predictorNum] - sign)
                    val = i32(val * sign)
                    errorVal = i32(errorVal - i32(sar(val, predictorQuantitization) * (predictorCoefNum - predictorNum)))
                    predictorNum -= 1
            bufferOutIdx += 1
        return bufferOut

    # ---- :338-367 -----------------------------------------------------------------------------------------------------------
    def Deinterlace16(self, bufferA, bufferB, bufferOut, numchannels, numsamples, interlacingShift, interlacingLeftweight):
        if numsamples <= 0:
            return
        if 0 != interlacingLeftweight:
            for i in range(numsamples):
                midright = bufferA[i]
                difference = bufferB[i]
                right = i32(midright - sar(i32(difference * interlacingLeftweight), interlacingShift))
                left = i32(right + difference)
                bufferOut[i * numchannels] = left
                bufferOut[i * numchannels + 1] = right
            return
        for i in range(numsamples):
            left = bufferA[i]
            right = bufferB[i]
            bufferOut[i * numchannels] = left
            bufferOut[i * numchannels + 1] = right

    # ---- :369-421 -------------------------------------------------------------------------------------------------------------
    def Deinterlace24(self, bufferA, bufferB, uncompressedBytes, uncompressedBytesBufferA, uncompressedBytesBufferB, bufferOut,
                      numchannels, numsamples, interlacingShift, interlacingLeftweight):
        if numsamples <= 0:
            return
        for i in range(numsamples):
            if interlacingLeftweight != 0:
                midright = bufferA[i]
                difference = bufferB[i]
                right = i32(midright - sar(i32(difference * interlacingLeftweight), interlacingShift))
                left = i32(right + difference)
            else:
                left = bufferA[i]
                right = bufferB[i]
            if uncompressedBytes != 0:
                mask = i32(~((0xFFFFFFFF << ((uncompressedBytes * 8) & 31)) & 0xFFFFFFFF))
                left = shl(left, uncompressedBytes * 8)
                right = shl(right, uncompressedBytes * 8)
                left = left | (uncompressedBytesBufferA[i] & mask)
                right = right | (uncompressedBytesBufferB[i] & mask)
            bufferOut[i * numchannels * 3] = left & 0xFF
            bufferOut[i * numchannels * 3 + 1] = sar(left, 8) & 0xFF
            bufferOut[i * numchannels * 3 + 2] = sar(left, 16) & 0xFF
            bufferOut[i * numchannels * 3 + 3] = right & 0xFF
            bufferOut[i * numchannels * 3 + 4] = sar(right, 8) & 0xFF
            bufferOut[i * numchannels * 3 + 5] = sar(right, 16) & 0xFF

    # ---- :428-719 -----------------------------------------------------------------------------------------------------------------
    def _read_coefs(self, table, count):
        for i in range(count):
            tempPred = self.Readbits(16)
            if tempPred > 32767:
                tempPred = tempPred - 65536
            table[i] = tempPred

    def _read_raw_24(self):                                               # :513-520 / :680-691
        m = 1 << (24 - 1)
        audiobits = self.Readbits(16)
        audiobits = shl(audiobits, self._setinfoSampleSize - 16)
        audiobits = audiobits | self.Readbits(self._setinfoSampleSize - 16)
        x = audiobits & ((1 << 24) - 1)
        return i32((x ^ m) - m)

    def DecodeFrame(self, inbuffer, outbuffer):
        outputsamples = self._setinfoMaxSamplesPerFrame
        self._inputBuffer = inbuffer
        self._inputBufferBitaccumulator = 0
        self._ibIdx = 0
        channels = self.Readbits(3)
        outputsize = i32(outputsamples * self._bytespersample)
        ss = self._setinfoSampleSize
        histmult_of = lambda ricemodifier: ricemodifier * cs_div(self._setinfoRiceHistorymult, 4)   # noqa: E731
        kmask = i32(shl(1, self._setinfoRiceKmodifier) - 1)
        if channels == 0:                                                 # 1 channel, :437-576
            self.Readbits(4)
            self.Readbits(12)
            hassize = self.Readbits(1)
            uncompressedBytes = self.Readbits(2)
            isnotcompressed = self.Readbits(1)
            if hassize != 0:
                outputsamples = self.Readbits(32)
                outputsize = i32(outputsamples * self._bytespersample)
            readsamplesize = ss - (uncompressedBytes * 8)
            if isnotcompressed == 0:
                self.Readbits(8)
                self.Readbits(8)
                predictionType = self.Readbits(4)
                predictionQuantitization = self.Readbits(4)
                ricemodifier = self.Readbits(3)
                predictorCoefNum = self.Readbits(5)
                self._read_coefs(self._predictorCoefTable, predictorCoefNum)
                if uncompressedBytes != 0:
                    for i in range(outputsamples):
                        self._uncompressedBytesBufferA[i] = self.Readbits(uncompressedBytes * 8)
                self.EntropyRiceDecode(self._predicterrorBufferA, outputsamples, readsamplesize, self._setinfoRiceInitialhistory,
                                       self._setinfoRiceKmodifier, histmult_of(ricemodifier), kmask)
                if predictionType == 0:
                    self._outputsamplesBufferA = self.PredictorDecompressFirAdapt(
                        self._predicterrorBufferA, outputsamples, readsamplesize, self._predictorCoefTable, predictorCoefNum,
                        predictionQuantitization)
                # else: nothing (:488-496) -- the output below is whatever _outputsamplesBufferA held
            else:
                if ss <= 16:
                    for i in range(outputsamples):
                        audiobits = self.Readbits(ss)
                        bitsmove = 32 - ss
                        self._outputsamplesBufferA[i] = sar(shl(audiobits, bitsmove), bitsmove)
                else:
                    for i in range(outputsamples):
                        self._outputsamplesBufferA[i] = self._read_raw_24()
                uncompressedBytes = 0
            if ss == 16:
                for i in range(outputsamples):
                    sample = self._outputsamplesBufferA[i]
                    outbuffer[i * self._numchannels] = sample
                    outbuffer[(i * self._numchannels) + 1] = 0
            elif ss == 24:
                for i in range(outputsamples):
                    sample = self._outputsamplesBufferA[i]
                    if uncompressedBytes != 0:
                        sample = shl(sample, uncompressedBytes * 8)
                        mask = i32(~((0xFFFFFFFF << ((uncompressedBytes * 8) & 31)) & 0xFFFFFFFF))
                        sample = sample | (self._uncompressedBytesBufferA[i] & mask)
                    nc3 = i * self._numchannels * 3
                    outbuffer[nc3] = sample & 0xFF
                    outbuffer[nc3 + 1] = sar(sample, 8) & 0xFF
                    outbuffer[nc3 + 2] = sar(sample, 16) & 0xFF
                    outbuffer[nc3 + 3] = 0
                    outbuffer[nc3 + 4] = 0
                    outbuffer[nc3 + 5] = 0
            else:
                raise Exception("FIXME: unimplemented sample size " + str(ss))
        elif channels == 1:                                               # 2 channels, :577-717
            self.Readbits(4)
            self.Readbits(12)
            hassize = self.Readbits(1)
            uncompressedBytes = self.Readbits(2)
            isnotcompressed = self.Readbits(1)
            if hassize != 0:
                outputsamples = self.Readbits(32)
                outputsize = i32(outputsamples * self._bytespersample)
            readsamplesize = ss - (uncompressedBytes * 8) + 1
            if isnotcompressed == 0:
                interlacingShift = self.Readbits(8)
                interlacingLeftweight = self.Readbits(8)
                predictionTypeA = self.Readbits(4)
                predictionQuantitizationA = self.Readbits(4)
                ricemodifierA = self.Readbits(3)
                predictorCoefNumA = self.Readbits(5)
                self._read_coefs(self._predictorCoefTableA, predictorCoefNumA)
                predictionTypeB = self.Readbits(4)
                predictionQuantitizationB = self.Readbits(4)
                ricemodifierB = self.Readbits(3)
                predictorCoefNumB = self.Readbits(5)
                self._read_coefs(self._predictorCoefTableB, predictorCoefNumB)
                if uncompressedBytes != 0:
                    for i in range(outputsamples):
                        self._uncompressedBytesBufferA[i] = self.Readbits(uncompressedBytes * 8)
                        self._uncompressedBytesBufferB[i] = self.Readbits(uncompressedBytes * 8)
                self.EntropyRiceDecode(self._predicterrorBufferA, outputsamples, readsamplesize, self._setinfoRiceInitialhistory,
                                       self._setinfoRiceKmodifier, histmult_of(ricemodifierA), kmask)
                if predictionTypeA == 0:
                    self._outputsamplesBufferA = self.PredictorDecompressFirAdapt(
                        self._predicterrorBufferA, outputsamples, readsamplesize, self._predictorCoefTableA, predictorCoefNumA,
                        predictionQuantitizationA)
                else:
                    raise Exception("FIXME: unhandled predicition type: " + str(predictionTypeA))
                self.EntropyRiceDecode(self._predicterrorBufferB, outputsamples, readsamplesize, self._setinfoRiceInitialhistory,
                                       self._setinfoRiceKmodifier, histmult_of(ricemodifierB), kmask)
                if predictionTypeB == 0:
                    self._outputsamplesBufferB = self.PredictorDecompressFirAdapt(
                        self._predicterrorBufferB, outputsamples, readsamplesize, self._predictorCoefTableB, predictorCoefNumB,
                        predictionQuantitizationB)
                else:
                    raise Exception("FIXME: unhandled predicition type: " + str(predictionTypeB))
            else:
                if ss <= 16:
                    for i in range(outputsamples):
                        audiobitsA = self.Readbits(ss)
                        audiobitsB = self.Readbits(ss)
                        bitsmove = 32 - ss
                        self._outputsamplesBufferA[i] = sar(shl(audiobitsA, bitsmove), bitsmove)
                        self._outputsamplesBufferB[i] = sar(shl(audiobitsB, bitsmove), bitsmove)
                else:
                    for i in range(outputsamples):
                        a = self._read_raw_24()
                        b = self._read_raw_24()
                        self._outputsamplesBufferA[i] = a
                        self._outputsamplesBufferB[i] = b
                uncompressedBytes = 0
                interlacingShift = 0
                interlacingLeftweight = 0
            if ss == 16:
                self.Deinterlace16(self._outputsamplesBufferA, self._outputsamplesBufferB, outbuffer, self._numchannels,
                                   outputsamples, interlacingShift, interlacingLeftweight)
            elif ss == 24:
                self.Deinterlace24(self._outputsamplesBufferA, self._outputsamplesBufferB, uncompressedBytes,
                                   self._uncompressedBytesBufferA, self._uncompressedBytesBufferB, outbuffer, self._numchannels,
                                   outputsamples, interlacingShift, interlacingLeftweight)
            elif ss in (20, 32):
                raise Exception("FIXME: unimplemented sample size " + str(ss))
            # any other sample size: nothing is written (:701-716)
        return outputsize


def decode_packet(cfg, packet, read_buffer_bytes=1024 * 80, out_ints=1024 * 80):
    """One packet through a FRESH decoder, the way AlacContext drives it (AlacContext.cs:54-55, :64-65, :195-197):
    the packet sits at the start of a zero-filled 80 KiB read buffer.  cfg = (max_samples_per_frame, sample_size,
    rice_history_mult, rice_initial_history, rice_kmodifier, num_channels).  Returns (outbuffer list, return value) or
    raises what the reference raises."""
    frame_len, sample_size, pb, mb, kb, nch = cfg
    f = AlacFile(sample_size, nch)
    cd = [0] * 24 + [(frame_len >> 24) & 255, (frame_len >> 16) & 255, (frame_len >> 8) & 255, frame_len & 255, 0, sample_size, pb,
                     mb, kb, nch, 0, 255, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xAC, 0x44]
    f.SetInfo(cd)
    buf = bytearray(read_buffer_bytes)
    buf[:len(packet)] = packet
    out = new_int_array(out_ints)
    ret = f.DecodeFrame(buf, out)
    return out, ret


def canonical_from_reference_layout(out, n, sample_size, nch):
    """the reference's int[] (24-bit: one int per byte) -> one int32 per sample, sign-extended (the build's canonical form)"""
    if sample_size != 24:
        return [out[i] for i in range(n * nch)]
    res = []
    for i in range(n * nch):
        v = out[3 * i] | (out[3 * i + 1] << 8) | (out[3 * i + 2] << 16)
        res.append(v - (1 << 24) if v & (1 << 23) else v)
    return res
