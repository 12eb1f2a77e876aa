"""On-disk database layout of `pyskani_amd.Database`.

File NAMES and write/flush semantics are the reference's (lib.rs:49-123, 187-227):
    separated     <folder>/<name>.sketch  written by sketch();   markers.bin written by flush()
    consolidated  <folder>/sketches.db    appended by sketch();  markers.bin + index.db written by flush()
The BYTES are this build's own versioned format, not skani's bincode: the serde layout of
skani::types::Sketch is defined by the absent crate and no sample database exists to read. Files written
here are therefore NOT interchangeable with the skani CLI / pyskani (README.md:105 of the reference claims
that interop for pyskani itself).

record  := magic "PSKS" u32 version | i32 c, marker_c, k | u32 name_len, name utf-8 | u32 n_contigs, u32 lens[]
           | u64 n_seeds, {u32 kmer, pos, contig, canon}[] | u64 n_markers, u64 markers[]      (little endian)
markers.bin := magic "PSKM" u32 version | i32 c, marker_c, k | u32 n | record[] with n_seeds = 0
index.db    := magic "PSKI" u32 version | u32 n | { u32 name_len, name, u64 offset, u64 length }[] sorted by offset
"""
import io
import os
import struct

import numpy as np

VERSION = 1
SEED_DTYPE = np.dtype([("kmer", "<u4"), ("pos", "<u4"), ("contig", "<u4"), ("canon", "<u4")])


class Record:
    __slots__ = ("params", "name", "contig_lens", "seeds", "markers")

    def __init__(self, params, name, contig_lens, seeds, markers):
        self.params = tuple(int(x) for x in params)           # (c, marker_c, k)
        self.name = name
        self.contig_lens = np.ascontiguousarray(contig_lens, dtype="<u4")
        self.seeds = np.ascontiguousarray(seeds, dtype=SEED_DTYPE)
        self.markers = np.ascontiguousarray(markers, dtype="<u8")

    def markers_only(self):
        return Record(self.params, self.name, self.contig_lens, np.zeros(0, SEED_DTYPE), self.markers)

    def to_bytes(self):
        name = self.name.encode("utf-8")
        out = io.BytesIO()
        out.write(b"PSKS" + struct.pack("<I3iI", VERSION, *self.params, len(name)) + name)
        out.write(struct.pack("<I", len(self.contig_lens)) + self.contig_lens.tobytes())
        out.write(struct.pack("<Q", len(self.seeds)) + self.seeds.tobytes())
        out.write(struct.pack("<Q", len(self.markers)) + self.markers.tobytes())
        return out.getvalue()

    @classmethod
    def read(cls, f):
        head = f.read(4 + 4 + 12 + 4)
        if len(head) != 24 or head[:4] != b"PSKS":
            raise ValueError("not a pyskani_amd sketch record")
        version, c, mc, k, nlen = struct.unpack("<I3iI", head[4:])
        if version != VERSION:
            raise ValueError(f"unsupported sketch record version {version}")
        name = f.read(nlen).decode("utf-8")
        (nc,) = struct.unpack("<I", f.read(4))
        lens = np.frombuffer(f.read(4 * nc), dtype="<u4")
        (ns,) = struct.unpack("<Q", f.read(8))
        seeds = np.frombuffer(f.read(16 * ns), dtype=SEED_DTYPE)
        (nm,) = struct.unpack("<Q", f.read(8))
        markers = np.frombuffer(f.read(8 * nm), dtype="<u8")
        if len(lens) != nc or len(seeds) != ns or len(markers) != nm:
            raise ValueError("truncated sketch record")
        return cls((c, mc, k), name, lens, seeds, markers)

    @classmethod
    def from_bytes(cls, data):
        return cls.read(io.BytesIO(data))


def write_markers(path, params, records):
    with open(path, "wb") as f:
        f.write(b"PSKM" + struct.pack("<I3iI", VERSION, *[int(x) for x in params], len(records)))
        for r in records:
            f.write(r.markers_only().to_bytes())


def read_markers(path):
    with open(path, "rb") as f:
        head = f.read(4 + 4 + 12 + 4)
        if len(head) != 24 or head[:4] != b"PSKM":
            raise ValueError(f"{path}: not a pyskani_amd marker file (skani/pyskani bincode databases cannot be read)")
        version, c, mc, k, n = struct.unpack("<I3iI", head[4:])
        if version != VERSION:
            raise ValueError(f"unsupported marker file version {version}")
        return (c, mc, k), [Record.read(f) for _ in range(n)]


def write_index(path, index):
    entries = sorted(index.items(), key=lambda kv: kv[1][0])      # by offset, lib.rs:207-208
    with open(path, "wb") as f:
        f.write(b"PSKI" + struct.pack("<II", VERSION, len(entries)))
        for name, (offset, length) in entries:
            b = name.encode("utf-8")
            f.write(struct.pack("<I", len(b)) + b + struct.pack("<QQ", offset, length))


def read_index(path):
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) != 12 or head[:4] != b"PSKI":
            raise ValueError(f"{path}: not a pyskani_amd index file")
        version, n = struct.unpack("<II", head[4:])
        if version != VERSION:
            raise ValueError(f"unsupported index file version {version}")
        index = {}
        for _ in range(n):
            (nlen,) = struct.unpack("<I", f.read(4))
            name = f.read(nlen).decode("utf-8")
            offset, length = struct.unpack("<QQ", f.read(16))
            index[name] = (offset, length)
        return index


class Folder:
    """DatabaseStorage::Folder (lib.rs:57-62, 99-107): one `<name>.sketch` per genome, silently overwritten."""
    kind = "separated"

    def __init__(self, path):
        self.path = path

    def store(self, record):
        with open(os.path.join(self.path, f"{record.name}.sketch"), "wb") as f:
            f.write(record.to_bytes())

    def load(self, name):
        with open(os.path.join(self.path, f"{name}.sketch"), "rb") as f:
            return Record.read(f)

    def flush(self, params, records):
        write_markers(os.path.join(self.path, "markers.bin"), params, records)


class Consolidated:
    """DatabaseStorage::Consolidated (lib.rs:64-87, 108-122): records appended to sketches.db + an index."""
    kind = "consolidated"

    def __init__(self, path, index=None, file_name="sketches.db"):
        self.path = path
        self.index = {} if index is None else index
        self.file_name = file_name

    def store(self, record):
        if record.name in self.index:                              # lib.rs:66-72
            raise ValueError(f"duplicate name in sketches: {record.name!r}")
        data = record.to_bytes()
        with open(os.path.join(self.path, self.file_name), "ab") as f:
            f.seek(0, os.SEEK_END)
            offset = f.tell()
            f.write(data)
        self.index[record.name] = (offset, len(data))

    def load(self, name):
        if name not in self.index:
            raise KeyError(name)                                   # lib.rs:109-112
        offset, length = self.index[name]
        with open(os.path.join(self.path, self.file_name), "rb") as f:
            f.seek(offset)
            return Record.from_bytes(f.read(length))

    def flush(self, params, records):
        write_markers(os.path.join(self.path, "markers.bin"), params, records)
        write_index(os.path.join(self.path, "index.db"), self.index)
