"""Minimal NetCDF-4 (HDF5) reader for the reference's input files - no netCDF4 / h5py / xarray needed.

The reference reads `experiments/data/**/*.nc` with `xr.open_dataarray(path)`
(experiments/pre_fitting_cmip6models.py:43,59; full_experiment_script.py:20,51,67): one variable `tas`
with dims (realisation, time) in float64, a `time` coordinate with CF `units` / `calendar` attributes and a
`realisation` coordinate (labels).  Those files are HDF5 with a version-0 superblock, contiguous,
uncompressed datasets; this module parses exactly that subset of the HDF5 file format (superblock 0-3,
object headers v1/v2 with continuation blocks, old-style groups through the v1 B-tree + local heap and
new-style groups through compact link messages, dataspace / datatype / contiguous-or-compact layout /
attribute messages, chunked datasets through the v1 chunk B-tree with optional deflate / shuffle / fletcher32
filters - the unlimited `time` dimension is stored in one-element chunks -, and dense attribute storage through
the fractal heap + v2 B-tree, which the observation file's `time` variable needs for its `units`) and raises
`NotImplementedError` for anything else (dense link storage, filtered heaps, variable-length data other than the
skipped coordinate labels).

    da = open_dataarray("CCCma_CanESM5-p1_ssp126_gmst.nc")      # labelled.DataArray, dims (realisation, time)
"""
from __future__ import annotations

import re
import zlib

import numpy as np

from .labelled import DataArray

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        base = 0
        while self.b[base:base + 8] != _SIG:
            base = 512 if base == 0 else base * 2
            if base >= len(self.b):
                raise ValueError("not an HDF5 / NetCDF-4 file")
        ver = self.b[base + 8]
        if ver in (0, 1):
            self.so, self.sl = self.b[base + 13], self.b[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = self._u(p, self.so)
            p += 4 * self.so                       # base, free-space, end-of-file, driver-info addresses
            # root group symbol-table entry: link name offset, object header address, cache type, scratch
            self.root = self._u(p + self.so, self.so)
        elif ver in (2, 3):
            self.so, self.sl = self.b[base + 9], self.b[base + 10]
            p = base + 12
            self.base = self._u(p, self.so)
            self.root = self._u(p + 3 * self.so, self.so)
        else:
            raise NotImplementedError(f"HDF5 superblock version {ver}")
        if self.base not in (0, _UNDEF & ((1 << (8 * self.so)) - 1)):
            raise NotImplementedError("non-zero HDF5 base address")

    # -- primitives ---------------------------------------------------------------------------------
    def _u(self, p, n):
        return int.from_bytes(self.b[p:p + n], "little")

    # -- object headers -----------------------------------------------------------------------------
    def messages(self, addr):
        """[(type, flags, payload bytes)] of the object header at addr (v1 or v2, continuations followed)."""
        b = self.b
        out = []
        if b[addr:addr + 4] == b"OHDR":
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16                               # four time stamps
            if flags & 0x10:
                p += 4                                # max compact / min dense attributes
            n = 1 << (flags & 3)
            size = self._u(p, n)
            p += n
            blocks = [(p, size)]
            track = bool(flags & 0x04)
            while blocks:
                q, sz = blocks.pop(0)
                end = q + sz
                while q + 4 <= end:
                    mtype, msize, mflags = b[q], self._u(q + 1, 2), b[q + 3]
                    q += 4 + (2 if track else 0)
                    payload = b[q:q + msize]
                    q += msize
                    if mtype == 0x10:                 # continuation: offset, length of an OCHK block
                        off, ln = self._u(q - msize, self.so), self._u(q - msize + self.so, self.sl)
                        blocks.append((off + 4, ln - 8))      # skip "OCHK", drop the trailing checksum
                    elif mtype != 0:
                        out.append((mtype, mflags, payload))
            return out
        if b[addr] != 1:
            raise NotImplementedError("unknown object header version")
        nmsg = self._u(addr + 2, 2)
        size = self._u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        while blocks and len(out) < nmsg + 64:
            q, sz = blocks.pop(0)
            end = q + sz
            while q + 8 <= end:
                mtype, msize, mflags = self._u(q, 2), self._u(q + 2, 2), b[q + 4]
                payload = b[q + 8:q + 8 + msize]
                if mtype == 0x10:
                    blocks.append((self._u(q + 8, self.so), self._u(q + 8 + self.so, self.sl)))
                elif mtype != 0:
                    out.append((mtype, mflags, payload))
                q += 8 + msize
        return out

    # -- groups --------------------------------------------------------------------------------------
    def links(self, addr):
        """{name: object header address} of the group at addr."""
        out = {}
        for mtype, _, m in self.messages(addr):
            if mtype == 0x06:                         # link message (new-style group, compact storage)
                flags = m[1]
                p = 2
                ltype = 0
                if flags & 0x08:
                    ltype = m[p]; p += 1
                if flags & 0x04:
                    p += 8                            # creation order
                if flags & 0x10:
                    p += 1                            # character set
                n = 1 << (flags & 3)
                ln = int.from_bytes(m[p:p + n], "little"); p += n
                name = m[p:p + ln].decode(); p += ln
                if ltype == 0:
                    out[name] = int.from_bytes(m[p:p + self.so], "little")
            elif mtype == 0x02:                       # link info: dense storage lives in a fractal heap
                flags = m[1]
                p = 2 + (8 if flags & 1 else 0)
                heap = int.from_bytes(m[p:p + self.so], "little")
                if heap != (1 << (8 * self.so)) - 1:
                    raise NotImplementedError("HDF5 group with dense link storage")
            elif mtype == 0x11:                       # symbol table (old-style group): B-tree + local heap
                btree = int.from_bytes(m[:self.so], "little")
                heap = int.from_bytes(m[self.so:2 * self.so], "little")
                out.update(self._symbol_table(btree, heap))
        return out

    def _symbol_table(self, btree, heap):
        b = self.b
        assert b[heap:heap + 4] == b"HEAP"
        data = self._u(heap + 8 + 2 * self.sl, self.so)
        out = {}

        def name_at(off):
            e = b.index(b"\0", data + off)
            return b[data + off:e].decode()

        def walk(node):
            if b[node:node + 4] == b"SNOD":
                n = self._u(node + 6, 2)
                p = node + 8
                for _ in range(n):
                    out[name_at(self._u(p, self.so))] = self._u(p + self.so, self.so)
                    p += 2 * self.so + 4 + 4 + 16
                return
            assert b[node:node + 4] == b"TREE" and b[node + 4] == 0
            n = self._u(node + 6, 2)
            p = node + 8 + 2 * self.so                # left / right siblings
            p += self.sl                              # key 0
            for _ in range(n):
                walk(self._u(p, self.so))
                p += self.so + self.sl
        walk(btree)
        return out

    # -- datasets ------------------------------------------------------------------------------------
    @staticmethod
    def _dtype(m):
        cls, ver = m[0] & 0x0F, m[0] >> 4
        bits0 = m[1]
        size = int.from_bytes(m[4:8], "little")
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            return np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}")
        if cls == 1:
            return np.dtype(f"{order}f{size}")
        if cls == 3:
            return np.dtype(f"S{size}")
        return None                                   # variable-length / compound / reference ...: not decoded

    def _shape(self, m):
        ver, rank = m[0], m[1]
        p = 8 if ver == 1 else 4
        return tuple(int.from_bytes(m[p + i * self.sl:p + (i + 1) * self.sl], "little") for i in range(rank))

    def _attribute(self, m):
        """(name, value) of an attribute message body (versions 1-3); value None for types that are not decoded."""
        ver = m[0]
        nlen, tlen, slen = (int.from_bytes(m[2 + 2 * i:4 + 2 * i], "little") for i in range(3))
        p = 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = m[p:p + nlen].split(b"\0")[0].decode(); p += pad(nlen)
        dt = self._dtype(m[p:p + tlen]); p += pad(tlen)
        space = m[p:p + slen]; p += pad(slen)
        shape = self._shape(space) if space[1] > 0 else ()
        if dt is None:
            return name, None
        count = int(np.prod(shape)) if shape else 1
        val = np.frombuffer(m, dtype=dt, count=count, offset=p)
        if dt.kind == "S":
            return name, val[0].split(b"\0")[0].decode() if count == 1 else [v.decode() for v in val]
        return name, val.reshape(shape) if shape else val[0]

    # -- dense attribute storage: fractal heap objects indexed by a v2 B-tree (attribute info message) -------
    def _dense_attributes(self, heap_addr, btree_addr):
        b, O, L = self.b, self.so, self.sl
        if b[heap_addr:heap_addr + 4] != b"FRHP":
            raise NotImplementedError("fractal heap header not found")
        p = heap_addr + 5
        p += 2                                        # heap id length
        filt_len = self._u(p, 2); p += 2
        hflags = b[p]; p += 1
        max_obj = self._u(p, 4); p += 4
        p += L + O + L + O + 4 * L + 4 * L            # huge-object / free-space / managed-space bookkeeping
        width = self._u(p, 2); p += 2
        start_size = self._u(p, L); p += L
        max_direct = self._u(p, L); p += L
        max_heap_bits = self._u(p, 2); p += 2
        p += 2                                        # starting rows of the root indirect block
        root = self._u(p, O); p += O
        cur_rows = self._u(p, 2)
        if filt_len:
            raise NotImplementedError("filtered fractal heap")
        off_bytes = (max_heap_bits + 7) // 8
        len_bytes = (min(max_direct, max_obj).bit_length() + 7) // 8
        blocks = []                                   # (heap offset of the block, file address, size)
        if cur_rows == 0:
            blocks.append((0, root, start_size))
        else:
            if b[root:root + 4] != b"FHIB":
                raise NotImplementedError("fractal heap indirect block not found")
            q = root + 5 + O + off_bytes
            hoff = 0
            for row in range(cur_rows):
                size = start_size if row < 2 else start_size << (row - 1)
                if size > max_direct:
                    raise NotImplementedError("fractal heap with nested indirect blocks")
                for _ in range(width):
                    child = self._u(q, O); q += O
                    if child != (1 << (8 * O)) - 1:
                        blocks.append((hoff, child, size))
                    hoff += size

        def heap_object(hid):
            if (hid[0] >> 4) & 3 != 0:
                raise NotImplementedError("huge / tiny fractal heap object")
            off = int.from_bytes(hid[1:1 + off_bytes], "little")
            ln = int.from_bytes(hid[1 + off_bytes:1 + off_bytes + len_bytes], "little")
            for hoff, addr, size in blocks:
                if hoff <= off < hoff + size:
                    return b[addr + off - hoff:addr + off - hoff + ln]
            raise ValueError("fractal heap offset outside every direct block")
        del hflags
        if b[btree_addr:btree_addr + 4] != b"BTHD":
            raise NotImplementedError("v2 B-tree header not found")
        rec_size = self._u(btree_addr + 10, 2)
        depth = self._u(btree_addr + 12, 2)
        node = self._u(btree_addr + 16, O)
        nrec = self._u(btree_addr + 16 + O, 2)
        if depth != 0 or b[node:node + 4] != b"BTLF":
            raise NotImplementedError("v2 B-tree deeper than one leaf")
        out = {}
        for i in range(nrec):
            rec = b[node + 6 + i * rec_size:node + 6 + (i + 1) * rec_size]
            try:
                k, v = self._attribute(heap_object(rec[:8]))
                out[k] = v
            except (NotImplementedError, ValueError):
                raise
            except Exception:
                pass
        return out

    # -- chunked datasets: v1 B-tree of chunks, optional deflate / shuffle / fletcher32 ------------------
    def _chunked(self, layout, shape, dt, filters):
        rank1 = layout[2]
        btree = int.from_bytes(layout[3:3 + self.so], "little")
        cdims = [int.from_bytes(layout[3 + self.so + 4 * i:7 + self.so + 4 * i], "little") for i in range(rank1)]
        chunk = tuple(cdims[:-1])
        out = np.zeros(shape, dtype=dt.newbyteorder("="))
        b = self.b

        def walk(node):
            assert b[node:node + 4] == b"TREE" and b[node + 4] == 1
            level, n = b[node + 5], self._u(node + 6, 2)
            p = node + 8 + 2 * self.so
            for _ in range(n):
                nbytes, mask = self._u(p, 4), self._u(p + 4, 4)
                offs = [self._u(p + 8 + 8 * i, 8) for i in range(rank1)]
                child = self._u(p + 8 + 8 * rank1, self.so)
                p += 8 + 8 * rank1 + self.so
                if level > 0:
                    walk(child)
                    continue
                raw = b[child:child + nbytes]
                for k, (fid, cd) in reversed(list(enumerate(filters))):
                    if mask & (1 << k):
                        continue
                    if fid == 3:
                        raw = raw[:-4]
                    elif fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        es = cd[0] if cd else dt.itemsize
                        raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
                    else:
                        raise NotImplementedError(f"HDF5 filter {fid}")
                block = np.frombuffer(raw, dtype=dt, count=int(np.prod(chunk))).reshape(chunk)
                sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, shape))
                out[sel] = block[tuple(slice(0, x.stop - x.start) for x in sel)]
        if btree != (1 << (8 * self.so)) - 1:
            walk(btree)
        return out

    def dataset(self, addr, want_data=True):
        shape, dt, layout, attrs, filters = (), None, None, {}, []
        for mtype, _, m in self.messages(addr):
            if mtype == 0x01:
                shape = self._shape(m)
            elif mtype == 0x03:
                dt = self._dtype(m)
            elif mtype == 0x08:
                layout = m
            elif mtype == 0x0B:                        # filter pipeline
                ver, nf = m[0], m[1]
                p = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid = int.from_bytes(m[p:p + 2], "little")
                    nlen = int.from_bytes(m[p + 2:p + 4], "little") if (ver == 1 or fid >= 256) else 0
                    p += 4 if (ver == 1 or fid >= 256) else 2
                    ncd = int.from_bytes(m[p + 2:p + 4], "little")
                    p += 4
                    p += (nlen + 7) & ~7 if ver == 1 else nlen
                    cd = [int.from_bytes(m[p + 4 * i:p + 4 * i + 4], "little") for i in range(ncd)]
                    p += 4 * ncd + (4 if ver == 1 and ncd % 2 else 0)
                    filters.append((fid, cd))
            elif mtype == 0x15:                        # attribute info: dense storage when the heap address is set
                flags = m[1]
                p = 2 + (2 if flags & 1 else 0)
                heap = int.from_bytes(m[p:p + self.so], "little")
                bt = int.from_bytes(m[p + self.so:p + 2 * self.so], "little")
                if heap != (1 << (8 * self.so)) - 1:
                    attrs.update(self._dense_attributes(heap, bt))
            elif mtype == 0x0C:
                try:
                    k, v = self._attribute(m)
                    attrs[k] = v
                except Exception:                     # an attribute this subset cannot decode is not fatal
                    pass
        data = None
        if want_data and dt is not None and layout is not None and dt.kind != "S":
            ver, cls = layout[0], layout[1]
            if ver not in (3, 4):
                raise NotImplementedError(f"HDF5 data layout message version {ver}")
            count = int(np.prod(shape)) if shape else 1
            if cls == 1:
                off = int.from_bytes(layout[2:2 + self.so], "little")
                data = np.frombuffer(self.b, dtype=dt, count=count, offset=off).reshape(shape)
            elif cls == 0:
                sz = int.from_bytes(layout[2:4], "little")
                data = np.frombuffer(layout[4:4 + sz], dtype=dt, count=count).reshape(shape)
            elif cls == 2 and ver == 3:
                data = self._chunked(layout, shape, dt, filters)
            else:
                raise NotImplementedError("HDF5 data layout class / version this reader does not know")
            data = data.astype(dt.newbyteorder("="), copy=True)
        return {"shape": shape, "dtype": dt, "attrs": attrs, "data": data}


_UNITS = re.compile(r"^\s*(days|hours|minutes|seconds)\s+since\s+(\d{1,4})-(\d{1,2})-(\d{1,2})(?:[T\s]+(\d{1,2}):(\d{1,2})(?::(\d{1,2}(?:\.\d*)?))?)?")
_STEP = {"days": 86400.0, "hours": 3600.0, "minutes": 60.0, "seconds": 1.0}


def decode_cf_time(values, units: str, calendar: str = "standard"):
    """CF time coordinate -> numpy datetime64[ns] for the (proleptic) Gregorian calendars the reference's files use."""
    cal = (calendar or "standard").lower()
    if cal not in ("standard", "gregorian", "proleptic_gregorian"):
        raise NotImplementedError(f"calendar {calendar!r}")
    mt = _UNITS.match(units)
    if not mt:
        raise ValueError(f"cannot parse time units {units!r}")
    unit, y, mo, d, hh, mm, ss = mt.groups()
    origin = np.datetime64(f"{int(y):04d}-{int(mo):02d}-{int(d):02d}", "s")
    origin = origin + np.timedelta64(int(hh or 0) * 3600 + int(mm or 0) * 60 + int(float(ss or 0)), "s")
    secs = np.round(np.asarray(values, dtype=np.float64) * _STEP[unit] * 1e3).astype("int64")    # milliseconds
    return (origin.astype("datetime64[ms]") + secs.astype("timedelta64[ms]")).astype("datetime64[ns]")


def open_dataarray(path, variable: str = None) -> DataArray:
    """xr.open_dataarray for the reference's data files: the single data variable with its decoded coordinates."""
    with open(path, "rb") as f:
        rd = _Reader(f.read())
    links = rd.links(rd.root)
    meta = {name: rd.dataset(addr, want_data=False) for name, addr in links.items()}
    dims_like = {n for n, m in meta.items() if m["attrs"].get("CLASS") == "DIMENSION_SCALE"}
    if variable is None:
        cand = [n for n in meta if n not in dims_like]
        if len(cand) != 1:
            raise ValueError(f"expected exactly one data variable, found {sorted(cand)}")
        variable = cand[0]
    var = rd.dataset(links[variable])
    if var["data"] is None:
        raise NotImplementedError(f"variable {variable!r} has a data type this reader does not decode")
    # NetCDF-4 ties dimensions to variables through object references (DIMENSION_LIST); the reference's files have
    # one coordinate per dimension with a distinct length, which identifies them without following references
    dims, coords = [], {}
    for ax, n in enumerate(var["shape"]):
        match = [d for d in sorted(dims_like) if meta[d]["shape"] == (n,) and d not in dims]
        name = match[0] if match else f"dim_{ax}"
        dims.append(name)
        if match:
            c = rd.dataset(links[name])
            if c["data"] is None:                      # e.g. variable-length string labels: positions instead
                coords[name] = np.arange(n)
            elif "units" in c["attrs"] and " since " in str(c["attrs"]["units"]):
                coords[name] = decode_cf_time(c["data"], c["attrs"]["units"], c["attrs"].get("calendar", "standard"))
            else:
                coords[name] = c["data"]
        else:
            coords[name] = np.arange(n)
    data = np.array(var["data"], dtype=np.float64)
    fill = var["attrs"].get("_FillValue")
    if fill is not None and np.isfinite(np.asarray(fill)).all():
        data[data == np.asarray(fill).reshape(-1)[0]] = np.nan
    return DataArray(data, tuple(dims), coords, variable)
