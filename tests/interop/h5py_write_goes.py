"""Run by an INDEPENDENT python (h5py; nothing of this repository is imported): writes a GOES-R L1b look-alike the way netCDF-4 files are laid
out on HDF5 -- coordinate variables as dimension scales (make_scale), data variables with their scales attached (attach_scale: DIMENSION_LIST /
REFERENCE_LIST), _Netcdf4Dimid, fixed-length string attributes, a deflate-compressed chunked Rad -- as netCDF4 / h5netcdf writers produce it.
The repository's reader (nc4lite, oct_goesread) then has to read a file it did not write.
usage: h5py_write_goes.py out.nc nx ny rad.bin(int16) t band xoff yoff"""
import sys

import h5py
import numpy as np

path, nx, ny, radbin, t, band, xoff, yoff = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], float(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7]), float(sys.argv[8])
rad = np.fromfile(radbin, dtype=np.int16).reshape(ny, nx)


def text(s):
    return np.string_(s)           # fixed-length string attribute, as netCDF-4 writes NC_CHAR attributes


with h5py.File(path, "w", libver="earliest", track_order=True) as f:
    y = f.create_dataset("y", data=np.arange(ny, dtype=np.int16))
    x = f.create_dataset("x", data=np.arange(nx, dtype=np.int16))
    y.make_scale("y"); x.make_scale("x")
    y.attrs["_Netcdf4Dimid"] = np.int32(0); x.attrs["_Netcdf4Dimid"] = np.int32(1)
    y.attrs["scale_factor"] = np.float32(-5.6e-05); y.attrs["add_offset"] = np.float32(yoff)
    x.attrs["scale_factor"] = np.float32(5.6e-05); x.attrs["add_offset"] = np.float32(xoff)
    r = f.create_dataset("Rad", data=rad, chunks=(min(ny, 32), min(nx, 64)), compression="gzip", compression_opts=4)
    r.dims[0].attach_scale(y); r.dims[1].attach_scale(x)
    r.attrs["scale_factor"] = np.float32(0.04572892); r.attrs["add_offset"] = np.float32(-1.6443)
    r.attrs["long_name"] = text("ABI L1b Radiances")
    tt = f.create_dataset("t", data=np.float64(t))
    tt.attrs["units"] = text("seconds since 2000-01-01 12:00:00")
    bd = f.create_dataset("band", data=np.zeros(1, np.float32))           # a dimension without a coordinate variable of its own type: a bare scale
    bd.make_scale("This is a netCDF dimension but not a netCDF variable.         1")
    bd.attrs["_Netcdf4Dimid"] = np.int32(2)
    b = f.create_dataset("band_id", data=np.array([band], dtype=np.int8))
    b.dims[0].attach_scale(bd)
    g = f.create_dataset("goes_imager_projection", data=np.int32(-2147483647))
    g.attrs["grid_mapping_name"] = text("geostationary")
    for k, v in (("perspective_point_height", 35786023.0), ("semi_major_axis", 6378137.0), ("semi_minor_axis", 6356752.31414),
                 ("inverse_flattening", 298.2572221), ("latitude_of_projection_origin", 0.0), ("longitude_of_projection_origin", -75.0)):
        g.attrs[k] = np.float64(v)
    g.attrs["sweep_angle_axis"] = text("x")
    for k, v in (("planck_fk1", 10803.3), ("planck_fk2", 1392.74), ("planck_bc1", 0.07550), ("planck_bc2", 0.99975), ("kappa0", 0.0015839)):
        f.create_dataset(k, data=np.float32(v))
