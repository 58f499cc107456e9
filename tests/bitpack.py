"""MSB-first bit packer for hand-written known-answer packets (tests only)."""


def pack(fields, slack=0):
    """fields: iterable of (nbits, value) or a '0101' string.  Zero-padded to a byte + `slack` bytes."""
    bits = []
    for f in fields:
        if isinstance(f, str):
            bits.extend(int(c) for c in f if c in "01")
        else:
            nbits, value = f
            value &= (1 << nbits) - 1
            bits.extend((value >> (nbits - 1 - i)) & 1 for i in range(nbits))
    while len(bits) % 8:
        bits.append(0)
    out = bytearray()
    for i in range(0, len(bits), 8):
        b = 0
        for j in range(8):
            b = (b << 1) | bits[i + j]
        out.append(b)
    return bytes(out) + bytes(slack)
