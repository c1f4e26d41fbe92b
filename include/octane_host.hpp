// octane_host.hpp -- C++ entry points with the reference's own signatures, implemented in
// octane_amd/csrc/host_shim.cpp on top of the C-ABI (octane_vof.h).  Linking liboctane_host.so in place of
// the reference's oct_variational_optical_flow.o / oct_pix2uv_cuda.o / oct_optical_flow.o is the drop-in.
#pragma once
#include <string>
#include <vector>

#include "octane_types.hpp"

// ref src/oct_variational_optical_flow.cu:1213 (declared by the caller at src/oct_optical_flow.cc:12).
// `nc` is ignored exactly as in the reference (it uses geo1i.nchannels, .cu:1223); CTH is never dereferenced
// (dodiscrete is hard-wired false, .cu:1302).  uarr/varr: first guess in, flow out.  Prints the reference's
// messages and exit(0)s when no GPU is present (.cu:1255-1259).
void oct_variational_optical_flow(Image geo1i, Image geo2i, float *CTH, float *uarr, float *varr,
                                  int nx, int ny, int nc, OFFlags args);

// ref src/oct_patch_match_optical_flow.cc:56 (declared at src/oct_optical_flow.cc:11): the -sosm method, a CPU loop in
// the reference, a HIP kernel here.  uarr/varr: first guess in (centres the search), displacement out.  Reads
// args.rad / args.srad.
void oct_patch_match_optical_flow(float *geo1i, float *geo2i, float *uarr, float *varr, int nx, int ny, OFFlags args);

// ref src/oct_pix2uv_cuda.cu:265 (declared at src/oct_optical_flow.cc:15).  Writes goesData.dT.
void oct_pix2uv_cuda(GOESVar &goesData, double t2, float *uarr, float *varr, short *ur, short *vr,
                     short *ur2, short *vr2, OFFlags args);

// ref src/oct_optical_flow.cc:21-111: zero / first-guess initialisation (oct_uv2pix), solver dispatch, CTP scaling,
// pix2uv, optional -srsal.  -sosm dispatches to oct_patch_match_optical_flow (one channel only, as in the reference).
int oct_optical_flow(GOESVar &goesData, GOESVar &goesData2, OFFlags &args);

// ref src/oct_pix2uv_cuda.cu:372: first-guess winds in u/v (m/s) -> pixel displacements, using goesData.latVal/lonVal/x/y
void oct_uv2pix(GOESVar &goesData, float *u, float *v, double t2, OFFlags args);
// ref src/oct_srsal_cuda.cu:73: bilateral smoothing of the flow, in place
void oct_srsal_cu(float *upix, float *vpix, float *CTHsub21, int nx, int ny, OFFlags args);

// ref src/oct_navcal_cuda.cu:100 (called by the GOES reader, src/oct_fileread.cc): raw counts -> calibrated, navigated,
// 0..255-normalised image of the window [minx,maxx) x [miny,maxy).  cal is "RAW" | "TEMP" | "REF" | "BRIT".
void oct_navcal_cuda(short *data2, short *data2s, short *x, short *y, short *xs, short *ys, int nx, int ny,
                     int minx, int maxx, int miny, int maxy, float *data3, float *lat, float *lon, std::string cal,
                     int datf, float xScale, float xOffset, float yScale, float yOffset, float radScale,
                     float radOffset, float rpol, float req, float H, float lam0, float fk1, float fk2, float bc1,
                     float bc2, float kap1, float maxin, float minin, float maxout, float minout, int donav,
                     OFFlags args);
// ref src/oct_normalize_geo.cc:9: per-band radiance range; leaves the outputs untouched for an unknown band
void oct_bandminmax(int gb, float &maxch, float &minch);
// ref src/oct_polar_navcal_cuda.cu:64 / src/oct_merc_navcal_cuda.cu:52 (called by the polar / mercator readers,
// src/oct_fileread.cc:567,728): navigation of re-mapped float images; lon0 / lat1 in degrees.  The polar form writes
// channel `chan` (1-based) of data3, i.e. at offset (chan-1) * window size, as the reference does.
void oct_polar_navcal_cuda(float *data2, short *data2s, short *x, short *y, short *xs, short *ys, int nx, int ny,
                           int minx, int maxx, int miny, int maxy, float *data3, float *lat, float *lon, float xScale,
                           float xOffset, float yScale, float yOffset, float lon0, float lat1, float R, int donav,
                           int chan, OFFlags args);
void oct_merc_navcal_cuda(float *data2, short *data2s, short *x, short *y, short *xs, short *ys, int nx, int ny,
                          int minx, int maxx, int miny, int maxy, float *data3, float *lat, float *lon, float xScale,
                          float xOffset, float yScale, float yOffset, float lon0, float R, int donav, OFFlags args);

// ref src/oct_zoom.cc:12,51,180 (host code in the reference too): resampling of a calibrated channel onto the grid of
// channel 1, used by the readers for -ic21/-ic22/-ic31/-ic32.  zoom_out: Gaussian blur (sigma = 0.6 sqrt(1/factor^2 - 1))
// + bicubic decimation to (int)(n factor + 0.5) pixels, a plain copy for factor >= 0.999999; zoom_in: bicubic (interp
// == 1) or nearest up-sampling with the half-pixel shift.  Channel `cnum` of the output is written.
void oct_zoom_size(int nx, int ny, int &nxx, int &nyy, double factor);
void oct_zoom_out_float(float *image, float *imageout, int nx, int ny, double factor, int verb, int cnum);
void oct_zoom_in_float(float *flow, float *flowout, int nx, int ny, int nxx, int nyy, int cnum, int interp);

// The `octane` command line (ref src/main.cc:42-50 spellings, :53-108 defaults, :166-350 scan), including
// its quirks: -scsig squares its argument, -set_device is 1-based, -corn clears docorn, -cgiters is not parsed.
struct OctaneCommandLine {
    OFFlags args;
    std::string f1, f2, f1c, f2c, fc21, fc22, fc31, fc32, f1fg;
    std::string interploc = "./interpolation", outdir = "./";
    bool show_help = false;     // argc < 4 (ref src/main.cc:112)
};
void octane_default_flags(OFFlags &args);
OctaneCommandLine octane_parse_command_line(int argc, const char *const *argv);
