"""Kaldi table I/O for the data formats either side of the path: feature matrices produced by
`compute-fbank-feats | add-deltas` (`exp/wsj/write_hdf_dataset.sh:94-104`) and transcripts, which the reference converts
to Fuel HDF5 with `bin/kaldi2fuel.py:103-360` through the external `kaldi_io` package.  This module reads and writes the
same containers directly:

  * binary archives of float matrices / vectors:  `<key> \\0B FM \\4<int32 rows>\\4<int32 cols><rows*cols float32>`
    (`DM` = float64, `FV`/`DV` vectors); several objects back to back in one `.ark`;
  * script files (`.scp`):  `<key> <path>:<byte offset>` lines pointing behind the key of an archive entry;
  * text archives:  `<key> [\\n r0c0 r0c1 ...\\n ... ]` matrices and `<key> tok tok ...` token / integer lines.

Kaldi's compressed matrices (`CM*`) are refused loudly.  Formats stated from Kaldi's documented table layout — no Kaldi
file ships with the reference, so the test is writer <-> reader round trips (parity UNPINNED).
"""
import struct

import numpy

_DTYPES = {b"F": numpy.float32, b"D": numpy.float64}


def _read_key(fh):
    key = bytearray()
    while True:
        c = fh.read(1)
        if not c:
            return None if not key else key.decode("utf-8")
        if c in b" \t\n":
            if key:
                return key.decode("utf-8")
            continue
        key += c


def _read_int(fh):
    size = fh.read(1)
    if size != b"\x04":
        raise ValueError("Kaldi binary: expected a 4-byte integer marker, got %r" % size)
    return struct.unpack("<i", fh.read(4))[0]


def _read_binary_object(fh):
    tok = fh.read(3)                               # "FM ", "DM ", "FV ", "DV ", "CM " ...
    if tok[:1] == b"C":
        raise ValueError("compressed Kaldi matrices are not supported: copy-feats --compress=false first")
    if tok[:1] not in _DTYPES or tok[1:2] not in (b"M", b"V") or tok[2:3] != b" ":
        raise ValueError("unknown Kaldi binary object token %r" % tok)
    dt = numpy.dtype(_DTYPES[tok[:1]])
    if tok[1:2] == b"M":
        rows, cols = _read_int(fh), _read_int(fh)
        data = fh.read(rows * cols * dt.itemsize)
        return numpy.frombuffer(data, dtype=dt.newbyteorder("<"), count=rows * cols).reshape(rows, cols).astype(dt)
    n = _read_int(fh)
    return numpy.frombuffer(fh.read(n * dt.itemsize), dtype=dt.newbyteorder("<"), count=n).astype(dt)


def _read_text_matrix(fh):
    rows, cur, tok = [], [], bytearray()
    opened = False
    while True:
        c = fh.read(1)
        if not c:
            raise ValueError("Kaldi text matrix: unexpected end of file")
        if c in b" \t\n]":
            if tok:
                if bytes(tok) == b"[":
                    opened = True
                else:
                    cur.append(float(tok))
                tok = bytearray()
            if c == b"\n" and cur:
                rows.append(cur)
                cur = []
            if c == b"]":
                if cur:
                    rows.append(cur)
                if not opened:
                    raise ValueError("Kaldi text matrix: ']' before '['")
                fh.readline()
                return numpy.asarray(rows, dtype=numpy.float32).reshape(len(rows), -1)
            continue
        tok += c
        if bytes(tok) == b"[":
            opened = True
            tok = bytearray()


def _read_object(fh):
    head = fh.read(2)
    if head == b"\x00B":
        return _read_binary_object(fh)
    fh.seek(-len(head), 1)
    return _read_text_matrix(fh)


def read_mat_ark(path_or_file):
    """Yield (key, ndarray) for every matrix / vector of a Kaldi archive (binary or text)."""
    fh = open(path_or_file, "rb") if isinstance(path_or_file, str) else path_or_file
    try:
        while True:
            key = _read_key(fh)
            if key is None:
                return
            yield key, _read_object(fh)
    finally:
        if isinstance(path_or_file, str):
            fh.close()


def read_mat_scp(path):
    """Yield (key, ndarray) following a `.scp`: `<key> <archive>[:<offset>]` per line."""
    handles = {}
    try:
        with open(path) as scp:
            for line in scp:
                line = line.strip()
                if not line:
                    continue
                key, rx = line.split(None, 1)
                offset = None
                if ":" in rx and rx.rsplit(":", 1)[1].isdigit():
                    rx, off = rx.rsplit(":", 1)
                    offset = int(off)
                fh = handles.get(rx)
                if fh is None:
                    fh = handles[rx] = open(rx, "rb")
                if offset is None:
                    fh.seek(0)
                    _read_key(fh)
                else:
                    fh.seek(offset)
                yield key, _read_object(fh)
    finally:
        for fh in handles.values():
            fh.close()


def write_mat_ark(path, items, scp=None):
    """Write (key, matrix-or-vector) pairs as a binary archive; optionally the matching `.scp`."""
    lines = []
    with open(path, "wb") as fh:
        for key, m in items:
            m = numpy.asarray(m)
            dt = b"D" if m.dtype == numpy.float64 else b"F"
            m = m.astype(numpy.float64 if dt == b"D" else numpy.float32)
            fh.write(key.encode("utf-8") + b" ")
            lines.append("%s %s:%d\n" % (key, path, fh.tell()))
            if m.ndim == 2:
                fh.write(b"\x00B" + dt + b"M " + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]))
            elif m.ndim == 1:
                fh.write(b"\x00B" + dt + b"V " + b"\x04" + struct.pack("<i", m.shape[0]))
            else:
                raise ValueError("Kaldi tables hold vectors and matrices only")
            fh.write(m.astype(m.dtype.newbyteorder("<")).tobytes())
    if scp:
        with open(scp, "w") as fh:
            fh.writelines(lines)


def read_text(path):
    """Kaldi `text` / integer-vector text archive: yield (key, [tokens])."""
    with open(path) as fh:
        for line in fh:
            p = line.split()
            if p:
                yield p[0], p[1:]


def compute_cmvn_stats(matrices):
    """Global mean / std over all frames (the reference's normalisation source, `lvsr/datasets/__init__.py:286-295`
    consumes a pickled (mean, std); `exp/wsj/write_hdf_dataset.sh:100-104` uses Kaldi's global CMVN)."""
    n, s, ss = 0, 0.0, 0.0
    for m in matrices:
        m = numpy.asarray(m, dtype=numpy.float64)
        n += m.shape[0]
        s = s + m.sum(0)
        ss = ss + (m * m).sum(0)
    mean = s / n
    return mean.astype(numpy.float32), numpy.sqrt(numpy.maximum(ss / n - mean * mean, 1e-20)).astype(numpy.float32)
