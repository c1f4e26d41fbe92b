"""Run by an INDEPENDENT python (this image: /opt/conda/bin/python3.9 with h5py 3.3 on HDF5 1.10.6 -- nothing of this repository is imported):
opens an outfile.nc written by octane_amd/csrc/io (nc4lite) with h5py, resolves every variable's dimensions through h5py's dimension-scale
interface (DIMENSION_LIST object references -> scale datasets; REFERENCE_LIST back-pointers checked by h5py's is_scale / iteration) -- the
mechanism netCDF-4 (and h5netcdf, which is built on exactly this interface) finds a variable's dimensions by -- and prints what it read as JSON.
usage: h5py_read_outfile.py file.nc"""
import json
import sys

import h5py
import numpy as np

out = {"vars": {}}
with h5py.File(sys.argv[1], "r") as f:
    for name, ds in f.items():
        if not isinstance(ds, h5py.Dataset):
            continue
        v = {"dtype": str(ds.dtype), "shape": list(ds.shape), "is_scale": bool(h5py.h5ds.is_scale(ds.id))}
        dims = []
        for d in ds.dims:
            dims.append([s.name.lstrip("/") for s in d.values()])        # the scales attached to this dimension
        v["dims"] = dims
        atts = {}
        for k, a in ds.attrs.items():
            if k in ("DIMENSION_LIST", "REFERENCE_LIST"):
                continue
            if isinstance(a, bytes):
                a = a.decode()
            elif isinstance(a, np.ndarray):
                a = a.tolist()              # netCDF attributes are 1-D arrays; one element is what ncdump prints as a scalar
                if isinstance(a, list) and len(a) == 1:
                    a = a[0]
            elif isinstance(a, np.generic):
                a = a.item()
            atts[k] = a
        v["atts"] = atts
        if ds.shape == () or ds.size <= 16:
            v["values"] = np.asarray(ds[()]).tolist()
        else:
            arr = np.asarray(ds[()])
            v["sum"] = float(arr.astype(np.float64).sum()); v["first"] = arr.ravel()[:4].tolist(); v["crc"] = int(np.bitwise_xor.reduce(arr.view(np.uint8).ravel().astype(np.uint32) * 2654435761 % (1 << 32)))
        out["vars"][name] = v
print(json.dumps(out))
