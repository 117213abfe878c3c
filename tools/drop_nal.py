#!/usr/bin/env python3
"""Writes a copy of an Annex-B bitstream without some of its NAL units (counted from 0), e.g. a coded picture and
the SEI message that follows it: a decoder then meets a reference picture that is missing and conceals it
(TDecTop::xCreateLostPicture of the reference decoder).

    drop_nal.py in.bin out.bin 7 8
"""
import re
import sys


def split_nals(data):
    starts = []
    for m in re.finditer(b"\x00\x00\x01", data):
        s = m.start()
        starts.append(s - 1 if s > 0 and data[s - 1] == 0 else s)
    return [data[s:(starts[i + 1] if i + 1 < len(starts) else len(data))] for i, s in enumerate(starts)]


def drop(data, which):
    return b"".join(n for i, n in enumerate(split_nals(data)) if i not in which)


if __name__ == "__main__":
    open(sys.argv[2], "wb").write(drop(open(sys.argv[1], "rb").read(), set(int(a) for a in sys.argv[3:])))
