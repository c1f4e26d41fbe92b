// host_shim.cpp -- the reference's C++ entry points (include/octane_host.hpp) on top of the C-ABI.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "../../include/octane_host.hpp"
#include "../../include/octane_vof.h"
#include "../../include/octane_extras.h"

// zero_guess: the caller KNOWS uarr / varr hold the zero first guess (oct_optical_flow below, without -firstguess): it is then not uploaded
static void variational_flow(Image geo1i, Image geo2i, float *uarr, float *varr, int nx, int ny, OFFlags args, bool zero_guess);

// OCTANE_HOST_TRACE=1: every entry point reports on stderr what arrived through the reference's by-value signature BEFORE anything touches
// the GPU -- what tests/test_ref_caller_link.py reads when the reference's own caller object (src/oct_optical_flow.cc, compiled where it
// lies with the reference's headers) is linked against this library on a machine without a GPU.
static bool host_trace() { const char *e = getenv("OCTANE_HOST_TRACE"); return e && *e && *e != '0'; }

void oct_variational_optical_flow(Image geo1i, Image geo2i, float *CTH, float *uarr, float *varr,
                                  int nx, int ny, int nc, OFFlags args)
{
    if (host_trace())
        fprintf(stderr, "TRACE oct_variational_optical_flow nx=%d ny=%d nc=%d geo1i={%d,%d,%d,%g} geo2i={%d,%d,%d,%g} u0=%g v0=%g "
                "args={alpha=%g lambda=%g lambdac=%g scaleF=%g scsig=%g kiters=%d liters=%d cgiters=%d dozim=%d setdevice=%d ftype=%s}\n",
                nx, ny, nc, geo1i.nrow, geo1i.ncol, geo1i.nchannels, geo1i.data ? geo1i.data[0] : 0.f, geo2i.nrow, geo2i.ncol, geo2i.nchannels,
                geo2i.data ? geo2i.data[0] : 0.f, uarr ? uarr[0] : 0.f, varr ? varr[0] : 0.f, args.alpha, args.lambda, args.lambdac, args.scaleF,
                args.scsig, args.kiters, args.liters, args.cgiters, args.dozim, args.setdevice, args.ftype.c_str());
    (void)CTH; (void)nc;         // CTH is never dereferenced (dodiscrete is hard-wired false, ref .cu:1302); nc is ignored (ref .cu:1223)
    variational_flow(geo1i, geo2i, uarr, varr, nx, ny, args, false);
}

static void variational_flow(Image geo1i, Image geo2i, float *uarr, float *varr, int nx, int ny, OFFlags args, bool zero_guess)
{
    (void)geo2i.nchannels;
    octane_vof_params p;
    octane_vof_default_params(&p);
    p.alpha = args.alpha; p.lambda = args.lambda; p.lambdac = args.lambdac;
    p.scaleF = args.scaleF; p.scsig = args.scsig;
    p.kiters = args.kiters; p.liters = args.liters; p.cgiters = args.cgiters;
    p.dozim = args.dozim; p.device = args.setdevice;
    const int ndev = octane_device_count();
    if (ndev == 0) {                         // ref .cu:1255-1259
        std::cout << "No gpus available for use, exiting\n";
        exit(0);
    }
    if (p.device > ndev - 1) {               // ref .cu:1260-1264
        std::cout << "Warning: setdevice set to non-existent GPU, setting to default GPU 1\n";
        p.device = 0;
    }
    // OCTANE_VOF_BANDS=n (2..8): solve this one frame on n GPUs as row bands, band b on device (setdevice + b) mod
    // the device count -- for frames like a 10848^2 full disk.  OFFlags has no field for it (the reference is
    // single-GPU), hence an environment variable; the result is the single-GPU one up to reduction order.
    int rc;
    const char *eb = getenv("OCTANE_VOF_BANDS");
    const int nbands = eb ? atoi(eb) : 1;
    if (nbands > 1) {
        int devs[8];
        for (int b = 0; b < nbands && b < 8; b++) devs[b] = (p.device + b) % ndev;
        octane_vof_tiled *t = nullptr;
        rc = octane_vof_tiled_create(&t, nx, ny, geo1i.nchannels, &p, nbands, devs, 0);
        if (rc == OCTANE_OK) rc = octane_vof_tiled_run(t, geo1i.data, geo2i.data, uarr, varr, OCTANE_MEM_HOST);
        octane_vof_tiled_destroy(t);
    } else {
        rc = octane_vof_solve(geo1i.data, geo2i.data, nx, ny, geo1i.nchannels, zero_guess ? nullptr : uarr, zero_guess ? nullptr : varr,
                              uarr, varr, &p);
    }
    if (rc != OCTANE_OK)   // the reference ignores CUDA errors (.cu:1421,1431); a failed solve is reported here
        std::cerr << "oct_variational_optical_flow: " << octane_last_error() << " (code " << rc << ")\n";
}

void oct_patch_match_optical_flow(float *geo1i, float *geo2i, float *uarr, float *varr, int nx, int ny, OFFlags args)
{
    const int ndev = octane_device_count();
    if (ndev == 0) {          // the reference runs this method on the CPU; here it needs the GPU like everything else
        std::cout << "No gpus available for use, exiting\n";
        exit(0);
    }
    const int dev = args.setdevice > ndev - 1 ? 0 : args.setdevice;
    const int rc = octane_sosm_run(geo1i, geo2i, nx, ny, uarr, varr, args.rad, args.srad, dev);
    if (rc != OCTANE_OK) std::cerr << "oct_patch_match_optical_flow: " << octane_last_error() << " (code " << rc << ")\n";
}

void oct_pix2uv_cuda(GOESVar &g, double t2, float *uarr, float *varr, short *ur, short *vr, short *ur2, short *vr2, OFFlags args)
{
    octane_nav nav;
    nav.pph = g.nav.pph; nav.req = g.nav.req; nav.rpol = g.nav.rpol; nav.lam0 = g.nav.lam0;
    nav.xScale = g.nav.xScale; nav.xOffset = g.nav.xOffset; nav.yScale = g.nav.yScale; nav.yOffset = g.nav.yOffset;
    nav.g2xOffset = g.nav.g2xOffset; nav.g2yOffset = g.nav.g2yOffset;
    nav.lat1 = g.nav.lat1; nav.lon1 = g.nav.lon1; nav.lon0 = g.nav.lon0; nav.R = g.nav.R;
    nav.minX = g.nav.minX; nav.minY = g.nav.minY; nav.nx = (int)g.nav.nx; nav.ny = (int)g.nav.ny;
    if (host_trace())
        fprintf(stderr, "TRACE oct_pix2uv_cuda t1=%g t2=%g nav={nx=%d ny=%d pph=%g xScale=%g yOffset=%g} pixuv=%d setdevice=%d\n", g.t, t2, nav.nx, nav.ny,
                nav.pph, (double)nav.xScale, (double)nav.yOffset, args.pixuv, args.setdevice);
    int dev = args.setdevice;
    if (args.pixuv == 0) {                   // the GPU is only needed for the navigated branch
        const int ndev = octane_device_count();
        if (ndev == 0) {
            std::cout << "No gpus available for use, exiting\n";
            exit(0);
        }
        if (dev > ndev - 1) {
            std::cout << "Warning: setdevice set to non-existent GPU, setting to default GPU 1\n";
            dev = 0;
        }
    }
    // Which build of the navigation kernel stands in for "the reference CUDA path": nvcc builds the reference's kernel with -fmad=true
    // (ref src/Makefile:9,20,27), and of the 13 multiply-add sites that may fuse only the two float ones of the base position move any
    // short (profiles/r5_pix2uv_sites.txt) -- the shim runs the strict build with exactly those two fused (include/octane_vof.h,
    // OCTANE_NAV_FMAD_FLOAT; OCTANE_PIX2UV_FMAD=0 in the environment selects the unfused build, for a reference built with -fmad=false).
    const int mode = (args.dopolar == 1 ? OCTANE_NAV_POLAR : (args.domerc == 1 ? OCTANE_NAV_MERC : OCTANE_NAV_GEOS)) | OCTANE_NAV_FMAD_FLOAT;
    float dT = 0.f;
    int moved = 0;
    const int rc = octane_pix2uv_run(&nav, g.t, t2, uarr, varr, args.pixuv, mode, ur, vr, ur2, vr2, &dT, &moved, dev);
    if (rc != OCTANE_OK) std::cerr << "oct_pix2uv_cuda: " << octane_last_error() << " (code " << rc << ")\n";
    if (moved)                                // ref p2u:359
        std::cout << "MOVE WARNING: Sector Moved, setting motions to 0 " << g.nav.xOffset << " " << g.nav.g2xOffset << " "
                  << g.nav.yOffset << " " << g.nav.g2yOffset << std::endl;
    g.dT = dT;
}

int oct_optical_flow(GOESVar &goesData, GOESVar &goesData2, OFFlags &args)
{
    const int nx = (int)goesData.nav.nx, ny = (int)goesData.nav.ny;
    const long n = (long)nx * ny;
    short *ur = new short[n], *vr = new short[n], *ur2 = new short[n], *vr2 = new short[n];
    if (args.dofirstguess == 0) {            // ref oct_optical_flow.cc:38-48
        goesData.uPix = new float[n];
        goesData.vPix = new float[n];
        for (long i = 0; i < n; i++) { goesData.uPix[i] = 0.f; goesData.vPix[i] = 0.f; }
    } else {                                 // first-guess file holds navigated winds: ref oct_optical_flow.cc:49-53
        oct_uv2pix(goesData, goesData.uPix, goesData.vPix, goesData2.t, args);
    }
    const int nc = 1 + args.doc2 + args.doc3;
    if (args.dososm == 1) {                   // ref oct_optical_flow.cc:57-64
        if (args.doc2 == 1 || args.doc3 == 1) {
            printf("Multichannel not yet supported on Patch Matching/Sum-of-Squared-error minimization, exiting\n");
            exit(0);
        }
        oct_patch_match_optical_flow(goesData.data.data, goesData2.data.data, goesData.uPix, goesData.vPix, nx, ny, args);
    } else {
        (void)nc;
        variational_flow(goesData.data, goesData2.data, goesData.uPix, goesData.vPix, nx, ny, args, args.dofirstguess == 0);
    }
    short *CTP = nullptr;
    if (args.doCTH == 1) {                   // ref oct_optical_flow.cc:71-88
        CTP = new short[n];
        for (long k = 0; k < n; k++)
            CTP[k] = (args.ir == 1) ? (short)((goesData.CTHVal[k] - 300) * 100) : (short)goesData.CTHVal[k];
    }
    oct_pix2uv_cuda(goesData, goesData2.t, goesData.uPix, goesData.vPix, ur, vr, ur2, vr2, args);
    goesData.uVal = ur; goesData.vVal = vr; goesData.uVal2 = ur2; goesData.vVal2 = vr2;
    if (args.dosrsal == 1) {                 // ref oct_optical_flow.cc:100-105
        std::cout << "Beginning anisotropic smoothing\n";
        oct_srsal_cu(goesData.uPix, goesData.vPix, goesData.CTHVal, nx, ny, args);
        std::cout << "Finished\n";
    }
    if (args.doCTH == 1) goesData.CTP = CTP;
    return 1;
}

static int pick_device(const OFFlags &args)
{
    const int ndev = octane_device_count();
    if (ndev == 0) {
        std::cout << "No gpus available for use, exiting\n";
        exit(0);
    }
    int dev = args.setdevice;
    if (dev > ndev - 1) {
        std::cout << "Warning: setdevice set to non-existent GPU, setting to default GPU 1\n";
        dev = 0;
    }
    return dev;
}

void oct_uv2pix(GOESVar &g, float *u, float *v, double t2, OFFlags args)
{
    octane_nav nav;
    nav.pph = g.nav.pph; nav.req = g.nav.req; nav.rpol = g.nav.rpol; nav.lam0 = g.nav.lam0;
    nav.xScale = g.nav.xScale; nav.xOffset = g.nav.xOffset; nav.yScale = g.nav.yScale; nav.yOffset = g.nav.yOffset;
    nav.g2xOffset = g.nav.g2xOffset; nav.g2yOffset = g.nav.g2yOffset;
    nav.lat1 = g.nav.lat1; nav.lon1 = g.nav.lon1; nav.lon0 = g.nav.lon0; nav.R = g.nav.R;
    nav.minX = g.nav.minX; nav.minY = g.nav.minY; nav.nx = (int)g.nav.nx; nav.ny = (int)g.nav.ny;
    const int rc = octane_uv2pix_run(&nav, g.t, t2, u, v, g.latVal, g.lonVal, g.x, g.y, pick_device(args));
    if (rc != OCTANE_OK) std::cerr << "oct_uv2pix: " << octane_last_error() << " (code " << rc << ")\n";
}

void oct_srsal_cu(float *upix, float *vpix, float *CTHsub21, int nx, int ny, OFFlags args)
{
    const int rc = octane_srsal_run(upix, vpix, CTHsub21, nx, ny, pick_device(args));
    if (rc != OCTANE_OK) std::cerr << "oct_srsal_cu: " << octane_last_error() << " (code " << rc << ")\n";
}

void oct_navcal_cuda(short *data2, short *data2s, short *x, short *y, short *xs, short *ys, int nx, int ny,
                     int minx, int maxx, int miny, int maxy, float *data3, float *lat, float *lon, std::string cal,
                     int datf, float xScale, float xOffset, float yScale, float yOffset, float radScale,
                     float radOffset, float rpol, float req, float H, float lam0, float fk1, float fk2, float bc1,
                     float bc2, float kap1, float maxin, float minin, float maxout, float minout, int donav,
                     OFFlags args)
{
    (void)datf;   // every calibration the reference accepts sets it to 1 (ref nav:52-75)
    octane_navcal_params p;
    p.xScale = xScale; p.xOffset = xOffset; p.yScale = yScale; p.yOffset = yOffset;
    p.radScale = radScale; p.radOffset = radOffset; p.rpol = rpol; p.req = req; p.H = H; p.lam0 = lam0;
    p.fk1 = fk1; p.fk2 = fk2; p.bc1 = bc1; p.bc2 = bc2; p.kap1 = kap1;
    p.maxin = maxin; p.minin = minin; p.maxout = maxout; p.minout = minout;
    p.cal = (cal == "TEMP") ? OCTANE_CAL_TEMP : (cal == "REF") ? OCTANE_CAL_REF : (cal == "BRIT") ? OCTANE_CAL_BRIT : OCTANE_CAL_RAW;
    p.donav = donav; p.minx = minx; p.maxx = maxx; p.miny = miny; p.maxy = maxy;
    const int ndev = octane_device_count();
    if (ndev == 0) {
        std::cout << "No gpus available for use, exiting\n";
        exit(0);
    }
    int dev = args.setdevice;
    if (dev > ndev - 1) {
        std::cout << "Warning: setdevice set to non-existent GPU, setting to default GPU 1\n";
        dev = 0;
    }
    const int rc = octane_navcal_run(data2, x, y, nx, ny, &p, data3, lat, lon, data2s, xs, ys, dev);
    if (rc != OCTANE_OK) std::cerr << "oct_navcal_cuda: " << octane_last_error() << " (code " << rc << ")\n";
}

static void proj_navcal(const char *who, int mode, float *data2, short *data2s, short *x, short *y, short *xs, short *ys,
                        int nx, int ny, int minx, int maxx, int miny, int maxy, float *data3, float *lat, float *lon,
                        float xScale, float xOffset, float yScale, float yOffset, float lon0, float lat1, float R, int donav,
                        const OFFlags &args)
{
    octane_proj_navcal_params p;
    p.xScale = xScale; p.xOffset = xOffset; p.yScale = yScale; p.yOffset = yOffset; p.lon0 = lon0; p.lat1 = lat1; p.R = R;
    p.donav = donav; p.mode = mode; p.minx = minx; p.maxx = maxx; p.miny = miny; p.maxy = maxy;
    const int ndev = octane_device_count();
    if (ndev == 0) {                              // ref pnav:84-88, mnav:69-73
        std::cout << "No gpus available for use, exiting\n";
        exit(0);
    }
    int dev = args.setdevice;
    if (dev > ndev - 1) {
        std::cout << "Warning: setdevice set to non-existent GPU, setting to default GPU 1\n";
        dev = 0;
    }
    const int rc = octane_proj_navcal_run(data2, x, y, nx, ny, &p, data3, lat, lon, data2s, xs, ys, dev);
    if (rc != OCTANE_OK) std::cerr << who << ": " << octane_last_error() << " (code " << rc << ")\n";
}

void oct_polar_navcal_cuda(float *data2, short *data2s, short *x, short *y, short *xs, short *ys, int nx, int ny,
                           int minx, int maxx, int miny, int maxy, float *data3, float *lat, float *lon, float xScale,
                           float xOffset, float yScale, float yOffset, float lon0, float lat1, float R, int donav,
                           int chan, OFFlags args)
{
    const long n2 = (long)(maxx - minx) * (maxy - miny);
    proj_navcal("oct_polar_navcal_cuda", OCTANE_NAV_POLAR, data2, data2s, x, y, xs, ys, nx, ny, minx, maxx, miny, maxy,
                data3 + (long)(chan - 1) * n2 /* ref pnav:140-143 */, lat, lon, xScale, xOffset, yScale, yOffset, lon0, lat1, R,
                donav, args);
}

void oct_merc_navcal_cuda(float *data2, short *data2s, short *x, short *y, short *xs, short *ys, int nx, int ny,
                          int minx, int maxx, int miny, int maxy, float *data3, float *lat, float *lon, float xScale,
                          float xOffset, float yScale, float yOffset, float lon0, float R, int donav, OFFlags args)
{
    proj_navcal("oct_merc_navcal_cuda", OCTANE_NAV_MERC, data2, data2s, x, y, xs, ys, nx, ny, minx, maxx, miny, maxy, data3,
                lat, lon, xScale, xOffset, yScale, yOffset, lon0, 0.f, R, donav, args);
}

void oct_bandminmax(int gb, float &maxch, float &minch)
{
    float mx, mn;
    if (octane_bandminmax(gb, &mx, &mn) == OCTANE_OK) { maxch = mx; minch = mn; }
}

void octane_default_flags(OFFlags &a)         // ref src/main.cc:53-108
{
    a.farn = 0; a.pixuv = 0; a.dosrsal = 0; a.dopolar = 0; a.domerc = 0; a.ftype = "GOES";
    a.fpyr_scale = 0.5f; a.flevels = 2; a.fwinsize = 20; a.fiterations = 5; a.poly_n = 10; a.poly_sigma = 0.5f;
    a.uif = 0; a.fg = 1; a.dofirstguess = 0; a.ir = 0; a.dososm = 0; a.dointerp = 0; a.docorn = 0;
    a.rad = 2; a.srad = 2; a.lambda = 1.; a.alpha = 5.; a.filtsigma = 3.; a.scaleF = 0.5; a.kiters = 4;
    a.alpha2 = 20.; a.lambdac = 0.; a.liters = 3; a.cgiters = 30; a.miters = 5; a.scsig = 400.;
    a.interpcth = 1; a.deltat = 60.f; a.doc2 = 0; a.doahi = 0; a.doc3 = 0; a.doinv = 0; a.doctt = 0; a.doCTH = 0;
    a.dozim = 1; a.outraw = true; a.outctp = true; a.outrad = true; a.outnav = true; a.setdevice = 0;
    a.setNormMax = true; a.setNormMin = true; a.setNormMax2 = true; a.setNormMin2 = true;
    a.setNormMax3 = true; a.setNormMin3 = true;
    // fields the reference leaves uninitialised are zeroed here
    a.putinterp = 0; a.oftype = 0; a.setnorms = 0;
    a.NormMax = a.NormMin = a.NormMax2 = a.NormMin2 = a.NormMax3 = a.NormMin3 = 0.f;
}

OctaneCommandLine octane_parse_command_line(int argc, const char *const *argv)
{
    OctaneCommandLine c;
    OFFlags &a = c.args;
    octane_default_flags(a);
    if (argc < 4) { c.show_help = true; return c; }
    // the reference reads argv[i+1] unguarded (src/main.cc:166-350); a trailing switch reads "" here instead
    auto val = [&](int i) -> const char * { return (i + 1 < argc) ? argv[i + 1] : ""; };
    for (int i = 0; i < argc; ++i) {
        const std::string s = argv[i];
        if (s == "-i1") c.f1 = val(i);
        if (s == "-i2") c.f2 = val(i);
        if (s == "-i1cth") { c.f1c = val(i); a.doCTH = 1; }
        if (s == "-i2cth") c.f2c = val(i);
        if (s == "-farn") { a.farn = 1; printf("Farneback disabled for this version of OCTANE, run without -farn, exiting..."); exit(0); }
        if (s == "-pd") a.pixuv = 1;
        if (s == "-srsal") a.dosrsal = 1;
        if (s == "-Polar") { a.dopolar = 1; a.ftype = "POLAR"; }
        if (s == "-Merc") { a.domerc = 1; a.ftype = "MERC"; }
        if (s == "-ahi") a.doahi = 1;
        if (s == "-ir") a.ir = 1;
        if (s == "-sosm") a.dososm = 1;
        if (s == "-interp") a.dointerp = 1;
        if (s == "-ic21") { a.doc2 = 1; c.fc21 = val(i); }
        if (s == "-ic22") c.fc22 = val(i);
        if (s == "-ic31") { a.doc3 = 1; c.fc31 = val(i); }
        if (s == "-ic32") c.fc32 = val(i);
        if (s == "-alpha") a.alpha = atof(val(i));
        if (s == "-lambda") a.lambda = atof(val(i));
        if (s == "-scsig") a.scsig = atof(val(i)) * atof(val(i));
        if (s == "-alpha2") a.alpha2 = atof(val(i));
        if (s == "-lambdac") a.lambdac = atof(val(i));
        if (s == "-fwinsize") a.fwinsize = atoi(val(i));
        if (s == "-polyn") a.poly_n = atoi(val(i));
        if (s == "-nncth") a.interpcth = 0;
        if (s == "-inv") a.doinv = 1;
        if (s == "-ctt") a.doctt = 1;
        if (s == "-kiters") a.kiters = atoi(val(i));
        if (s == "-liters") a.liters = atoi(val(i));
        if (s == "-brox") a.dozim = 0;
        if (s == "-corn") a.docorn = 0;
        if (s == "-firstguess") { a.dofirstguess = 1; c.f1fg = val(i); }
        if (s == "-rad") a.rad = atoi(val(i));
        if (s == "-srad") a.srad = atoi(val(i));
        if (s == "-deltat") a.deltat = (float)atof(val(i));
        if (s == "-interploc") c.interploc = val(i);
        if (s == "-no_outnav") a.outnav = false;
        if (s == "-no_outraw") a.outraw = false;
        if (s == "-no_outrad") a.outrad = false;
        if (s == "-no_outctp") a.outctp = false;
        if (s == "-set_device") a.setdevice = atoi(val(i)) - 1;
        if (s == "-normmax") { a.NormMax = (float)atof(val(i)); a.setNormMax = false; }
        if (s == "-normmin") { a.NormMin = (float)atof(val(i)); a.setNormMin = false; }
        if (s == "-normmax2") { a.NormMax2 = (float)atof(val(i)); a.setNormMax2 = false; }
        if (s == "-normmin2") { a.NormMin2 = (float)atof(val(i)); a.setNormMin2 = false; }
        if (s == "-normmax3") { a.NormMax3 = (float)atof(val(i)); a.setNormMax3 = false; }
        if (s == "-normmin3") { a.NormMin3 = (float)atof(val(i)); a.setNormMin3 = false; }
        if (s == "-o") c.outdir = val(i);
    }
    // ref src/main.cc:369-401
    a.oftype = (a.farn == 1) ? 2 : (a.dozim == 0 ? 3 : 1);
    if (a.dososm == 1) a.oftype = 4;
    if (a.dopolar == 1 || a.domerc == 1 || a.doahi == 1) a.doCTH = 0;
    return c;
}
