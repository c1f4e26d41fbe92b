// goes_io.cpp -- GOES-R L1b input and the `outfile.nc` output of the reference's CLI, on nc4lite.
//
// Behavioural spec: ref src/oct_fileread.cc:43-419 (oct_goesread) and :832-895 (oct_fileread dispatch);
// ref src/oct_filewrite.cc:17-349 (oct_goeswrite) and its dispatch.  Same C++ signatures, same GOESVar fields
// filled, same variable / attribute names written.  Only the GOES fixed-grid file type with one channel is
// provided; -Polar / -Merc / -ahi readers, cloud-top-height and first-guess files and the second / third channel
// (which need the CPU zoom helpers) are reported as unsupported.
//
// "parity unpinned" for this layer: the reference's I/O goes through netcdf-cxx4, which does not exist here, it has no
// tests or sample files, and no file written by it is available to compare against.
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "../../../include/octane_host.hpp"
#include "nc4lite.hpp"

static const int NC_ERR = 2;

int oct_goesread(std::string fpath, std::string cal, int donav, int channelnum, GOESVar &resVar, OFFlags &args)
{
    const double PI = 3.14159265359;
    const double DTOR = PI / 180.;
    if (channelnum != 1) {
        std::cout << "This build reads one channel per image only (-ic21/-ic22/-ic31/-ic32 need the CPU zoom helpers), exiting\n";
        exit(0);
    }
    try {
        nc4lite::Reader f(fpath);
        const long xdimsize = (long)f.dim_size("x"), ydimsize = (long)f.dim_size("y");     // ref fr:76-81
        const long nv = xdimsize * ydimsize;
        for (const char *v : {"Rad", "y", "x", "t", "band_id", "goes_imager_projection", "planck_fk1", "planck_fk2",
                              "planck_bc1", "planck_bc2", "kappa0"})
            if (!f.has_var(v)) return NC_ERR;
        GOESNAVVar &nav = resVar.nav;
        const float radScale = f.att_float("Rad", "scale_factor"), radOffset = f.att_float("Rad", "add_offset");
        nav.radScale = radScale; nav.radOffset = radOffset;                                  // ref fr:105-127
        const float yScale = f.att_float("y", "scale_factor"), yOffset = f.att_float("y", "add_offset");
        const float xScale = f.att_float("x", "scale_factor"), xOffset = f.att_float("x", "add_offset");
        nav.yScale = yScale; nav.yOffset = yOffset; nav.xScale = xScale; nav.xOffset = xOffset;
        resVar.tUnits = f.att_text("t", "units");
        {   // the reference reads the int variable's bytes into a float (fr:151); what it then writes back is the
            // same bytes, so the value is carried as an int through that float here too
            int gip = 0;
            f.read("goes_imager_projection", &gip);
            float asfloat;
            static_assert(sizeof(asfloat) == sizeof(gip), "int and float differ in size");
            std::memcpy(&asfloat, &gip, sizeof gip);
            nav.gipVal = asfloat;
        }
        const char *gp = "goes_imager_projection";
        nav.lpo = f.att_float(gp, "longitude_of_projection_origin");
        const float req = f.att_float(gp, "semi_major_axis"), rpol = f.att_float(gp, "semi_minor_axis");
        nav.req = req; nav.rpol = rpol;
        nav.inverse_flattening = f.att_float(gp, "inverse_flattening");
        nav.lat0 = f.att_float(gp, "latitude_of_projection_origin");
        const float pph = f.att_float(gp, "perspective_point_height");
        nav.pph = pph;
        float lam0 = f.att_float(gp, "longitude_of_projection_origin");
        lam0 = (float)(lam0 * DTOR);                                                          // ref fr:179-182
        nav.lam0 = lam0;
        float fk1, fk2, bc1, bc2, kap1;
        f.read("planck_fk1", &fk1); f.read("planck_fk2", &fk2); f.read("planck_bc1", &bc1); f.read("planck_bc2", &bc2);
        f.read("kappa0", &kap1);
        nav.fk1 = fk1; nav.fk2 = fk2; nav.bc1 = bc1; nav.bc2 = bc2; nav.kap1 = kap1;
        const float H = pph + req;                                                            // ref fr:263
        const int minx = 0, maxx = (int)xdimsize, miny = 0, maxy = (int)ydimsize;
        const int nc = 1 + (args.doc2 == 1) + (args.doc3 == 1);
        resVar.data.setdims(maxx - minx, maxy - miny, nc);
        resVar.data.data = new float[(size_t)(maxx - minx) * (maxy - miny) * nc];
        float *lat = new float[nv], *lon = new float[nv];
        short *xs = new short[maxx - minx], *ys = new short[maxy - miny], *data2s = new short[nv];
        short *data2 = new short[nv], *x = new short[xdimsize], *y = new short[ydimsize];
        nav.nx = maxx - minx; nav.ny = maxy - miny;
        f.read("y", y); f.read("x", x);
        f.read("t", &resVar.t);
        f.read("Rad", data2);
        int band = 0;
        f.read("band_id", &band);
        if (band == 2) { nav.minXc = minx / 4; nav.minYc = miny / 4; nav.maxXc = maxx / 4; nav.maxYc = maxy / 4; }       // ref fr:321-341
        else if (band == 1 || band == 3) { nav.minXc = minx / 2; nav.minYc = miny / 2; nav.maxXc = maxx / 2; nav.maxYc = maxy / 2; }
        else { nav.minXc = minx; nav.minYc = miny; nav.maxXc = maxx; nav.maxYc = maxy; }
        nav.minX = minx; nav.minY = miny; nav.maxX = maxx; nav.maxY = maxy;
        float maxch = 0.f, minch = 0.f;
        oct_bandminmax(band, maxch, minch);
        if (args.setNormMax) args.NormMax = maxch;
        if (args.setNormMin) args.NormMin = minch;
        oct_navcal_cuda(data2, data2s, x, y, xs, ys, (int)xdimsize, (int)ydimsize, minx, maxx, miny, maxy, resVar.data.data, lat, lon,
                        cal, 0, xScale, xOffset, yScale, yOffset, radScale, radOffset, rpol, req, H, lam0, fk1, fk2, bc1, bc2, kap1,
                        maxch, minch, 255.f, 0.f, donav, args);
        resVar.latVal = lat; resVar.lonVal = lon; resVar.x = xs; resVar.y = ys; resVar.dataSVal = data2s; resVar.band = band;
        delete[] data2; delete[] x; delete[] y;
    } catch (const nc4lite::Error &e) {
        std::cout << e.what() << "\n";
        std::cout << "OCT_GOESREAD FAILURE, CHECK THAT ALL VARIABLES AND ATTS EXIST" << std::endl;     // ref fr:411-415
        exit(1);
    }
    return 1;
}

// ref fr:818-858 oct_fgread: a first-guess file holds navigated winds UFG / VFG (m/s) on the image grid
int oct_fgread(std::string fpath, GOESVar &resVar, OFFlags &args)
{
    (void)args;
    try {
        nc4lite::Reader f(fpath);
        if (!f.has_var("UFG") || !f.has_var("VFG")) return NC_ERR;
        const std::vector<size_t> shp = f.shape("UFG");
        size_t nv = 1;
        for (size_t d : shp) nv *= d;
        if (nv != (size_t)resVar.nav.nx * (size_t)resVar.nav.ny) {
            std::cout << "First-guess file does not have the image's dimensions, exiting\n";
            exit(0);
        }
        float *u = new float[nv], *v = new float[nv];
        f.read("UFG", u); f.read("VFG", v);
        resVar.uPix = u; resVar.vPix = v;
    } catch (const nc4lite::Error &e) {
        std::cout << e.what() << "\nOCT_FGREAD FAILURE, CHECK THAT ALL VARIABLES AND ATTS EXIST" << std::endl;
        return NC_ERR;
    }
    return 1;
}

int oct_fileread(std::string fpath, std::string ftype, std::string cal, int donav, int channelnum, GOESVar &resVar, OFFlags &args)
{
    (void)cal;                                   // the reference's dispatcher passes "RAW" whatever it was given (fr:867)
    if (ftype == "GOES") return oct_goesread(fpath, "RAW", donav, channelnum, resVar, args);
    if (ftype == "FIRSTGUESS") return oct_fgread(fpath, resVar, args);
    std::cout << "File type " << ftype << " is not supported by this build (GOES fixed-grid L1b only), exiting\n";
    exit(0);
}

int oct_goeswrite(std::string fpath, GOESVar &resVar, OFFlags args)
{
    using nc4lite::Type;
    try {
        nc4lite::Writer w(fpath);
        const GOESNAVVar &nav = resVar.nav;
        w.def_dim("x", (size_t)nav.nx);
        w.def_dim("y", (size_t)nav.ny);
        w.def_var("x", Type::Short, {"x"});
        w.def_var("y", Type::Short, {"y"});
        w.put_att("x", "scale_factor", nav.xScale); w.put_att("x", "add_offset", nav.xOffset);
        w.put_att("y", "scale_factor", nav.yScale); w.put_att("y", "add_offset", nav.yOffset);
        w.put_var("x", resVar.x);
        w.put_var("y", resVar.y);
        w.def_var("t", Type::Double);
        w.put_att("t", "standard_name", std::string("time"));
        w.put_att("t", "units", resVar.tUnits);
        w.put_att("t", "axis", std::string("T"));
        w.put_att("t", "bounds", std::string("time_bounds"));
        w.put_att("t", "long_name", std::string("J2000 epoch mid-point between the start and end image scan in seconds"));
        if (args.putinterp == 1) w.put_att("t", "frdt", resVar.frdt);
        w.put_var("t", args.putinterp == 0 ? &resVar.t : &resVar.tint);

        const std::vector<std::string> yx = {"y", "x"};
        const std::string gm = "goes_imager_projection";
        auto flow_var = [&](const char *name, const char *long_name, const char *units) {
            w.def_var(name, Type::Short, yx, 1);
            w.put_att(name, "long_name", std::string(long_name));
            w.put_att(name, "grid_mapping", gm);
            w.put_att(name, "scale_factor", 0.01f);
            w.put_att(name, "units", std::string(units));
        };
        if (args.outnav) {                                             // ref fw:66-70,129-161
            flow_var("U", "U", args.pixuv == 1 ? "x-pixels" : "meters per second");
            flow_var("V", "V", args.pixuv == 1 ? "y-pixels" : "meters per second");
        }
        if (args.outraw) {
            flow_var("U_raw", "U Raw", "x-pixels");
            flow_var("V_raw", "V Raw", "y-pixels");
        }
        if (args.pixuv == 1) {
            for (const char *n : {"Upix", "Vpix"}) {
                w.def_var(n, Type::Float, yx, 1);
                w.put_att(n, "long_name", std::string(n));
                w.put_att(n, "grid_mapping", gm);
            }
        }
        if (args.outctp && args.doCTH == 1) {
            w.def_var("CTP", Type::Short, yx, 1);
            w.put_att("CTP", "long_name", std::string("CTP"));
            w.put_att("CTP", "grid_mapping", gm);
            w.put_att("CTP", "interpcth", (float)args.interpcth);
        }
        if (args.outrad) {
            w.def_var("Rad", Type::Short, yx, 1);
            w.put_att("Rad", "long_name", std::string("Rad"));
            w.put_att("Rad", "grid_mapping", gm);
            w.put_att("Rad", "scale_factor", nav.radScale);
            w.put_att("Rad", "add_offset", nav.radOffset);
        }
        w.def_var(gm, Type::Int);
        w.put_att(gm, "long_name", std::string("GOES-R ABI fixed grid projection"));
        w.put_att(gm, "grid_mapping_name", std::string("geostationary"));
        w.put_att(gm, "perspective_point_height", (double)nav.pph);
        w.put_att(gm, "semi_major_axis", (double)nav.req);
        w.put_att(gm, "semi_minor_axis", (double)nav.rpol);
        w.put_att(gm, "inverse_flattening", (double)nav.inverse_flattening);
        w.put_att(gm, "latitude_of_projection_origin", (double)nav.lat0);
        w.put_att(gm, "longitude_of_projection_origin", (double)nav.lpo);
        w.put_att(gm, "sweep_angle_axis", std::string("x"));
        {
            int gip;
            std::memcpy(&gip, &nav.gipVal, sizeof gip);               // see oct_goesread
            w.put_var(gm, &gip);
        }
        const char *of = "optical_flow_settings";
        w.def_var(of, Type::Int);
        w.put_att(of, "long_name", std::string("Optical Flow Settings"));
        w.put_att(of, "key", std::string("1 = Modified Zimmer et al. (2011), 2 = Farneback, 3 = Brox (2004), 4 = Least Squares"));
        w.put_att(of, "Image2_xOffset", nav.g2xOffset);
        w.put_att(of, "Image2_yOffset", nav.g2yOffset);
        if (args.oftype == 1 || args.oftype == 3) {                   // ref fw:237-251
            w.put_att(of, "lambda", args.lambda);
            w.put_att(of, "lambdac", args.lambdac);
            w.put_att(of, "alpha", args.alpha);
            w.put_att(of, "filtsigma", args.filtsigma);
            w.put_att(of, "ScaleF", args.scaleF);
            w.put_att(of, "K_Iterations", args.kiters);
            w.put_att(of, "L_Iterations", args.liters);
            w.put_att(of, "M_Iterations", args.miters);
            w.put_att(of, "CG_Iterations", args.cgiters);
            w.put_att(of, "NormMax", args.NormMax);
            w.put_att(of, "NormMin", args.NormMin);
            w.put_att(of, "dofirstguess", args.dofirstguess);
        } else if (args.oftype == 4) {                                // ref fw:268-273
            w.put_att(of, "Rad", args.rad);
            w.put_att(of, "SRad", args.srad);
            w.put_att(of, "NormMax", args.NormMax);
            w.put_att(of, "NormMin", args.NormMin);
        }
        w.put_att(of, "dt_seconds", resVar.dT);
        w.put_var(of, &args.oftype);            // the reference never writes a value (fill); its `key` attribute describes this one
        if (args.outnav) { w.put_var("U", resVar.uVal); w.put_var("V", resVar.vVal); }
        if (args.outraw) { w.put_var("U_raw", resVar.uVal2); w.put_var("V_raw", resVar.vVal2); }
        if (args.pixuv == 1) { w.put_var("Upix", resVar.uPix); w.put_var("Vpix", resVar.vPix); }
        if (args.outctp && args.doCTH == 1) w.put_var("CTP", resVar.CTP);
        if (args.outrad) {
            w.put_var("Rad", resVar.dataSVal);
            for (auto kv : {std::make_pair("planck_fk1", nav.fk1), std::make_pair("planck_fk2", nav.fk2), std::make_pair("planck_bc1", nav.bc1),
                            std::make_pair("planck_bc2", nav.bc2), std::make_pair("kappa0", nav.kap1)}) {
                w.def_var(kv.first, Type::Float);
                w.put_var(kv.first, &kv.second);
            }
        }
        w.close();
        return 0;
    } catch (const nc4lite::Error &e) {
        std::cout << e.what() << "\nGOESWRITE failure\n";
        return NC_ERR;
    }
}

int oct_filewrite(std::string fpath, std::string ftype, GOESVar &resVar, OFFlags args)
{
    if (ftype == "GOES") return oct_goeswrite(fpath, resVar, args);
    std::cout << "File type " << ftype << " is not supported by this build (GOES fixed-grid L1b only)\n";
    return NC_ERR;
}
