// goes_io.cpp -- the file layer of the reference's CLI on nc4lite: GOES-R L1b, polar-stereographic and
// Mercator re-mapped images, CLAVR-x cloud-top heights and first-guess winds in; outfile.nc / outfile_polar.nc /
// outfile_merc.nc out.
//
// Behavioural spec: ref src/oct_fileread.cc:43-419 (oct_goesread), :421-609 (oct_polarread), :611-754 (oct_mercread),
// :756-815 (oct_clavrxread), :817-857 (oct_fgread), :860-895 (oct_fileread); ref src/oct_filewrite.cc:17-349
// (oct_goeswrite), :353-563 (oct_polarwrite), :565-700 (oct_mercwrite) and its dispatch.  Same C++ signatures, same
// GOESVar fields filled, same variable / attribute names and types written.  Not provided: the -ahi reader and the
// frames of -interp (oct_interp is outside SURVEY 8).
//
// Where the reference reads uninitialised memory the sane value is used instead, and said so at the spot: the x / y
// scaling of a second or third GOES channel (never read from its file there, fr:128-183), `band` of the polar and
// Mercator readers (never assigned, fr:572,738).
//
// "parity unpinned" for this layer: the reference's I/O goes through netcdf-cxx4, which does not exist here, it has no
// tests or sample files, and no file written by it is available to compare against.  The channel resampling the readers
// call (zoom_host.cpp) is pinned bit for bit against the reference's own functions.
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/octane_host.hpp"
#include "nc4lite.hpp"

static const int NC_ERR = 2;

// Channel c lives in plane c - 1 of data.data and the buffer has one plane per requested channel: a third channel
// without a second one would be written past its end (the reference does exactly that, fr:279-283 with fr:373 / polar
// navcal's plane offset).  Refused instead.
static void need_second_channel_for_third(const OFFlags &args)
{
    if (args.doc3 == 1 && args.doc2 != 1) {
        std::cout << "A third channel (-ic31/-ic32) needs a second one (-ic21/-ic22), exiting\n";
        exit(0);
    }
}

int oct_goesread(std::string fpath, std::string cal, int donav, int channelnum, GOESVar &resVar, OFFlags &args)
{
    const double PI = 3.14159265359;
    const double DTOR = PI / 180.;
    if (channelnum < 1 || channelnum > 3) return NC_ERR;
    need_second_channel_for_third(args);
    try {
        nc4lite::Reader f(fpath);
        const long xdimsize = (long)f.dim_size("x"), ydimsize = (long)f.dim_size("y");     // ref fr:76-81
        const long nv = xdimsize * ydimsize;
        for (const char *v : {"Rad", "y", "x", "t", "band_id", "goes_imager_projection", "planck_fk1", "planck_fk2",
                              "planck_bc1", "planck_bc2", "kappa0"})
            if (!f.has_var(v)) return NC_ERR;
        GOESNAVVar &nav = resVar.nav;
        const float radScale = f.att_float("Rad", "scale_factor"), radOffset = f.att_float("Rad", "add_offset");
        if (channelnum == 1) { nav.radScale = radScale; nav.radOffset = radOffset; }       // ref fr:105-127
        if (channelnum == 2) { nav.radScale2 = radScale; nav.radOffset2 = radOffset; }
        if (channelnum == 3) { nav.radScale3 = radScale; nav.radOffset3 = radOffset; }
        // The reference reads the grid of channel 1 only and hands oct_navcal_cuda uninitialised scales for the others
        // (fr:128-183); every channel's own grid is read here, and only channel 1's is kept in `nav`.
        const float yScale = f.att_float("y", "scale_factor"), yOffset = f.att_float("y", "add_offset");
        const float xScale = f.att_float("x", "scale_factor"), xOffset = f.att_float("x", "add_offset");
        const char *gp = "goes_imager_projection";
        const float req = f.att_float(gp, "semi_major_axis"), rpol = f.att_float(gp, "semi_minor_axis");
        const float pph = f.att_float(gp, "perspective_point_height");
        float lam0 = f.att_float(gp, "longitude_of_projection_origin");
        lam0 = (float)(lam0 * DTOR);                                                          // ref fr:179-182
        if (channelnum == 1) {
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
            nav.lpo = f.att_float(gp, "longitude_of_projection_origin");
            nav.req = req; nav.rpol = rpol;
            nav.inverse_flattening = f.att_float(gp, "inverse_flattening");
            nav.lat0 = f.att_float(gp, "latitude_of_projection_origin");
            nav.pph = pph;
            nav.lam0 = lam0;
        }
        float fk1, fk2, bc1, bc2, kap1;
        f.read("planck_fk1", &fk1); f.read("planck_fk2", &fk2); f.read("planck_bc1", &bc1); f.read("planck_bc2", &bc2);
        f.read("kappa0", &kap1);
        if (channelnum == 1) { nav.fk1 = fk1; nav.fk2 = fk2; nav.bc1 = bc1; nav.bc2 = bc2; nav.kap1 = kap1; }      // ref fr:184-263
        if (channelnum == 2) { nav.fk12 = fk1; nav.fk22 = fk2; nav.bc12 = bc1; nav.bc22 = bc2; nav.kap12 = kap1; }
        if (channelnum == 3) { nav.fk13 = fk1; nav.fk23 = fk2; nav.bc13 = bc1; nav.bc23 = bc2; nav.kap13 = kap1; }
        const float H = pph + req;                                                            // ref fr:263
        const int minx = 0, maxx = (int)xdimsize, miny = 0, maxy = (int)ydimsize;
        const int wx = maxx - minx, wy = maxy - miny;
        float *data3 = nullptr;
        if (channelnum == 1) {
            const int nc = 1 + (args.doc2 == 1) + (args.doc3 == 1);
            resVar.data.setdims(wx, wy, nc);
            resVar.data.data = new float[(size_t)wx * wy * nc];
        } else {
            if (!resVar.data.data || nav.nx <= 0 || nav.ny <= 0) {
                std::cout << "Channel " << channelnum << " read before channel 1, exiting\n";
                exit(0);
            }
            data3 = new float[nv];                                                            // ref fr:284
        }
        float *lat = new float[nv], *lon = new float[nv];
        short *xs = new short[wx], *ys = new short[wy], *data2s = new short[nv];
        short *data2 = new short[nv], *x = new short[xdimsize], *y = new short[ydimsize];
        if (channelnum == 1) { nav.nx = wx; nav.ny = wy; }
        if (channelnum == 2) { nav.nx2 = wx; nav.ny2 = wy; }
        if (channelnum == 3) { nav.nx3 = wx; nav.ny3 = wy; }
        f.read("y", y); f.read("x", x);
        if (channelnum == 1) f.read("t", &resVar.t);
        f.read("Rad", data2);
        int band = 0;
        f.read("band_id", &band);
        if (channelnum == 1) {
            if (band == 2) { nav.minXc = minx / 4; nav.minYc = miny / 4; nav.maxXc = maxx / 4; nav.maxYc = maxy / 4; }       // ref fr:321-341
            else if (band == 1 || band == 3) { nav.minXc = minx / 2; nav.minYc = miny / 2; nav.maxXc = maxx / 2; nav.maxYc = maxy / 2; }
            else { nav.minXc = minx; nav.minYc = miny; nav.maxXc = maxx; nav.maxYc = maxy; }
        }
        nav.minX = minx; nav.minY = miny; nav.maxX = maxx; nav.maxY = maxy;
        float maxch = 0.f, minch = 0.f;
        oct_bandminmax(band, maxch, minch);
        if (channelnum == 1) { if (args.setNormMax) args.NormMax = maxch; if (args.setNormMin) args.NormMin = minch; }     // ref fr:350-364
        if (channelnum == 2) { if (args.setNormMax2) args.NormMax2 = maxch; if (args.setNormMin2) args.NormMin2 = minch; }
        if (channelnum == 3) { if (args.setNormMax3) args.NormMax3 = maxch; if (args.setNormMin3) args.NormMin3 = minch; }
        oct_navcal_cuda(data2, data2s, x, y, xs, ys, (int)xdimsize, (int)ydimsize, minx, maxx, miny, maxy,
                        channelnum == 1 ? resVar.data.data : data3, lat, lon,
                        cal, 0, xScale, xOffset, yScale, yOffset, radScale, radOffset, rpol, req, H, lam0, fk1, fk2, bc1, bc2, kap1,
                        maxch, minch, 255.f, 0.f, donav, args);
        if (channelnum == 1) {
            resVar.latVal = lat; resVar.lonVal = lon; resVar.x = xs; resVar.y = ys; resVar.dataSVal = data2s; resVar.band = band;
        } else {
            // onto channel 1's grid (ref fr:372-383): finer grids are blurred and decimated, coarser ones interpolated
            if (nav.nx > wx) {
                oct_zoom_in_float(data3, resVar.data.data, wx, wy, (int)nav.nx, (int)nav.ny, channelnum - 1, 1);
            } else {
                const double factor = (double)nav.nx / (double)wx, factor2 = (double)nav.ny / (double)wy;
                if (pow(factor - factor2, 2) > 0.000001) {
                    printf("Image x and y dimensions not compatable for scaling (factor not the same), exiting");
                    exit(0);
                }
                int zx, zy;
                oct_zoom_size(wx, wy, zx, zy, factor);
                if (zx != nav.nx || zy != nav.ny) {             // the reference would write a plane of another shape into the buffer
                    std::cout << "Channel " << channelnum << " does not scale onto channel 1's grid (" << zx << "x" << zy << " vs "
                              << nav.nx << "x" << nav.ny << "), exiting\n";
                    exit(0);
                }
                oct_zoom_out_float(data3, resVar.data.data, wx, wy, factor, 0, channelnum - 1);
            }
            if (channelnum == 2) resVar.band2 = band;
            if (channelnum == 3) resVar.band3 = band;
            delete[] data3; delete[] lat; delete[] lon; delete[] xs; delete[] ys; delete[] data2s;   // not kept (leaked by the reference)
        }
        delete[] data2; delete[] x; delete[] y;
    } catch (const nc4lite::Error &e) {
        std::cout << e.what() << "\n";
        std::cout << "OCT_GOESREAD FAILURE, CHECK THAT ALL VARIABLES AND ATTS EXIST" << std::endl;     // ref fr:411-415
        exit(1);
    }
    return 1;
}

// Re-mapped (polar-stereographic / Mercator) images: float `Rad` already calibrated, short x / y axes with scale and
// offset in metres, projection constants on the `grid_mapping` variable.  ref fr:421-609 and fr:611-754.
static int read_projected(const std::string &fpath, bool polar, int donav, int channelnum, GOESVar &resVar, OFFlags &args)
{
    try {
        nc4lite::Reader f(fpath);
        const long xdimsize = (long)f.dim_size("x"), ydimsize = (long)f.dim_size("y");
        const long nv = xdimsize * ydimsize;
        for (const char *v : {"Rad", "y", "x", "t", "grid_mapping"})
            if (!f.has_var(v)) return NC_ERR;
        GOESNAVVar &nav = resVar.nav;
        const float yScale = f.att_float("y", "scale_factor"), yOffset = f.att_float("y", "add_offset");
        const float xScale = f.att_float("x", "scale_factor"), xOffset = f.att_float("x", "add_offset");
        nav.yScale = yScale; nav.yOffset = yOffset; nav.xScale = xScale; nav.xOffset = xOffset;       // every channel, as in the reference
        resVar.tUnits = f.att_text("t", "units");
        int gipv = 0;
        f.read("grid_mapping", &gipv);
        nav.gipVal = (float)gipv;
        float lat1 = 0.f, lon0 = 0.f, lon1 = 0.f;
        if (polar) {
            lat1 = f.att_float("grid_mapping", "lat1"); lon0 = f.att_float("grid_mapping", "lon0");
            nav.lat1 = lat1; nav.lon0 = lon0;
        } else {
            lon1 = f.att_float("grid_mapping", "lon1");
            nav.lon1 = lon1;
        }
        const float R = f.att_float("grid_mapping", "R");
        nav.R = R;
        const int minx = 0, maxx = (int)xdimsize, miny = 0, maxy = (int)ydimsize;
        const int wx = maxx - minx, wy = maxy - miny;
        if (channelnum == 1) {
            const int nc = polar ? 1 + (args.doc2 == 1) + (args.doc3 == 1) : 1;
            resVar.data.setdims(wx, wy, nc);
            resVar.data.data = new float[(size_t)wx * wy * nc];
        } else if (!resVar.data.data || nav.nx != wx || nav.ny != wy) {
            std::cout << "Channel " << channelnum << " must have channel 1's grid (re-mapped files are not rescaled), exiting\n";
            exit(0);
        }
        float *lat = new float[nv], *lon = new float[nv];
        short *xs = new short[wx], *ys = new short[wy], *data2s = new short[nv];
        float *data2 = new float[nv];
        short *x = new short[xdimsize], *y = new short[ydimsize];
        if (channelnum == 1) { nav.nx = wx; nav.ny = wy; }
        if (channelnum == 2) { nav.nx2 = wx; nav.ny2 = wy; }
        if (channelnum == 3) { nav.nx3 = wx; nav.ny3 = wy; }
        f.read("y", y); f.read("x", x);
        if (channelnum == 1) f.read("t", &resVar.t);
        f.read("Rad", data2);
        if (channelnum == 1) { nav.minXc = minx; nav.minYc = miny; nav.maxXc = maxx; nav.maxYc = maxy; }
        nav.minX = minx; nav.minY = miny;
        if (polar)
            oct_polar_navcal_cuda(data2, data2s, x, y, xs, ys, (int)xdimsize, (int)ydimsize, minx, maxx, miny, maxy, resVar.data.data,
                                  lat, lon, xScale, xOffset, yScale, yOffset, lon0, lat1, R, donav, channelnum, args);
        else
            oct_merc_navcal_cuda(data2, data2s, x, y, xs, ys, (int)xdimsize, (int)ydimsize, minx, maxx, miny, maxy, resVar.data.data,
                                 lat, lon, xScale, xOffset, yScale, yOffset, lon1, R, donav, args);
        const size_t wn = (size_t)wx * wy;
        auto plane_copy = [&]() {                // -interp keeps a copy of the channel (ref fr:575-595 hands over an unfilled array)
            float *c = new float[wn];
            std::memcpy(c, resVar.data.data + (size_t)(channelnum - 1) * wn, wn * sizeof(float));
            return c;
        };
        if (channelnum == 1) {
            resVar.latVal = lat; resVar.lonVal = lon; resVar.x = xs; resVar.y = ys; resVar.dataSVal = data2s;
            resVar.band = 0;                     // never assigned in the reference (fr:572, fr:738)
            if (polar && args.dointerp == 1) resVar.dataSValfloat = plane_copy();
        } else {
            if (args.dointerp == 1) { if (channelnum == 2) resVar.dataSValfloat2 = plane_copy(); else resVar.dataSValfloat3 = plane_copy(); }
            delete[] lat; delete[] lon; delete[] xs; delete[] ys; delete[] data2s;
        }
        delete[] data2; delete[] x; delete[] y;
    } catch (const nc4lite::Error &e) {
        std::cout << e.what() << "\n";
        std::cout << (polar ? "OCT_POLARREAD" : "OCT_MERCREAD") << " FAILURE, CHECK THAT ALL VARIABLES AND ATTS EXIST" << std::endl;
        exit(1);
    }
    return 1;
}

int oct_polarread(std::string fpath, std::string cal, int donav, int channelnum, GOESVar &resVar, OFFlags &args)
{
    (void)cal;
    if (channelnum < 1 || channelnum > 3) return NC_ERR;
    need_second_channel_for_third(args);
    return read_projected(fpath, true, donav, channelnum, resVar, args);
}

int oct_mercread(std::string fpath, std::string cal, int donav, GOESVar &resVar, OFFlags &args)
{
    (void)cal;
    return read_projected(fpath, false, donav, 1, resVar, args);
}

// ref fr:756-815: CLAVR-x cloud-top heights (`Cloud_Top_Height_Effective` on dimensions nx, ny) brought onto the image
// grid; their window is channel 1's in CLAVR-x (2 km) coordinates, nav.minXc .. nav.maxXc
int oct_clavrxread(std::string fpath, GOESVar &resVar, OFFlags &args)
{
    try {
        nc4lite::Reader f(fpath);
        const char *name = "Cloud_Top_Height_Effective";
        if (!f.has_var(name)) return NC_ERR;
        GOESNAVVar &nav = resVar.nav;
        const int xmax = nav.maxXc, xmin = nav.minXc, ymax = nav.maxYc, ymin = nav.minYc;
        const int cx = xmax - xmin, cy = ymax - ymin;
        const std::vector<size_t> shp = f.shape(name);
        size_t nv = 1;
        for (size_t d : shp) nv *= d;
        if (cx <= 0 || cy <= 0 || nv < (size_t)cx * (size_t)cy) {
            std::cout << "Cloud-top height file is smaller than the image window, exiting\n";
            exit(0);
        }
        std::vector<float> data3(nv);
        f.read(name, data3.data());
        nav.CTHx = cx; nav.CTHy = cy;
        resVar.CTHVal = new float[(size_t)nav.nx * nav.ny];
        if (nav.nx > cx) {
            oct_zoom_in_float(data3.data(), resVar.CTHVal, cx, cy, (int)nav.nx, (int)nav.ny, 0, args.interpcth);
        } else {
            const double factor = (double)nav.nx / (double)cx, factor2 = (double)nav.ny / (double)cy;
            if (pow(factor - factor2, 2) > 0.000001) {
                printf("Image x and y dimensions not compatable for scaling (factor not the same), CTH data problem");
                exit(0);
            }
            oct_zoom_out_float(data3.data(), resVar.CTHVal, cx, cy, factor, 0, 0);
        }
    } catch (const nc4lite::Error &e) {
        std::cout << e.what() << "\nOCT_CLAVRXREAD FAILURE, CHECK THAT ALL VARIABLES AND ATTS EXIST" << std::endl;
        return NC_ERR;
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
    if (ftype == "POLAR") return oct_polarread(fpath, "RAW", donav, channelnum, resVar, args);
    if (ftype == "MERC") return oct_mercread(fpath, "RAW", donav, resVar, args);
    if (ftype == "CLAVRX") return oct_clavrxread(fpath, resVar, args);
    if (ftype == "FIRSTGUESS") return oct_fgread(fpath, resVar, args);
    std::cout << "File type " << ftype << " is not supported by this build, exiting\n";
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

// What oct_polarwrite and oct_mercwrite share: axes, time, the settings variable.  ref fw:353-563 / fw:565-700.
namespace {

void put_axes_and_time(nc4lite::Writer &w, GOESVar &resVar, const OFFlags &args, bool interp_time)
{
    using nc4lite::Type;
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
    if (interp_time && args.dointerp == 1) w.put_att("t", "frdt", resVar.frdt);
    w.put_var("t", (interp_time && args.putinterp != 0) ? &resVar.tint : &resVar.t);
}

void put_settings(nc4lite::Writer &w, const GOESVar &resVar, const OFFlags &args, bool brox_too)
{
    const char *of = "optical_flow_settings";
    w.def_var(of, nc4lite::Type::Int);
    w.put_att(of, "long_name", std::string("Optical Flow Settings"));
    w.put_att(of, "key", std::string("1 = Modified Sun (2014), 2 = Farneback, 3 = Brox (2004)"));
    if (args.oftype == 1 || (brox_too && args.oftype == 3)) {
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
    }
    // oftype 2 (Farneback, OpenCV) does not exist in this build, so its attributes are never written
    w.put_att(of, "dt_seconds", resVar.dT);
    w.put_var(of, &args.oftype);                 // the reference leaves the value unwritten (fill)
}

}  // namespace

// ref fw:353-563.  U / V are doubles holding the pixel displacements uPix / vPix (the reference hands the float arrays
// to a double variable); Rad, Rad2, Rad3 the normalised channels as floats.
int oct_polarwrite(std::string fpath, GOESVar &resVar, OFFlags args)
{
    using nc4lite::Type;
    try {
        nc4lite::Writer w(fpath);
        const GOESNAVVar &nav = resVar.nav;
        put_axes_and_time(w, resVar, args, true);
        const std::vector<std::string> yx = {"y", "x"};
        const std::string gm = "polar_orthonormal";
        for (const char *n : {"U", "V"}) {
            w.def_var(n, Type::Double, yx);
            w.put_att(n, "long_name", std::string(n));
            w.put_att(n, "grid_mapping", gm);
        }
        w.put_att("U", "units", std::string(args.pixuv == 0 ? "meters per second" : "x-pixels"));
        w.put_att("V", "units", std::string(args.pixuv == 1 ? "y-pixels" : "meters per second"));
        if (args.pixuv == 1) {
            for (const char *n : {"Upix", "Vpix"}) {
                w.def_var(n, Type::Float, yx);
                if (args.dosrsal == 1) w.put_att(n, "long_name", std::string(n));      // ref fw:428-432
            }
        }
        const int nrad = args.outrad ? 1 + (args.doc2 == 1) + (args.doc3 == 1) : 0;
        const char *radn[3] = {"Rad", "Rad2", "Rad3"};
        for (int c = 0, k = 0; c < 3 && k < nrad; c++) {
            if ((c == 1 && args.doc2 != 1) || (c == 2 && args.doc3 != 1)) continue;
            w.def_var(radn[c], Type::Float, yx);
            w.put_att(radn[c], "long_name", std::string(radn[c]));
            w.put_att(radn[c], "grid_mapping", gm);
            k++;
        }
        if (args.dointerp == 1) {
            w.def_var("Occlusion", Type::Short, yx);
            w.put_att("Occlusion", "long_name", std::string("Occlusion Masks"));
            w.put_att("Occlusion", "key", std::string("0 - both, 1 - only in image 1, 2 - only in image 2"));
        }
        const char *gip = "polar_imager_projection";
        w.def_var(gip, Type::Int);
        w.put_att(gip, "long_name", std::string("Polar_Orthonormal_Grid"));
        w.put_att(gip, "grid_mapping_name", std::string("polar"));
        w.put_att(gip, "lat1", (double)nav.lat1);
        w.put_att(gip, "lon0", (double)nav.lon0);
        w.put_att(gip, "R", (double)nav.R);
        put_settings(w, resVar, args, true);
        w.put_var("U", resVar.uPix);
        w.put_var("V", resVar.vPix);
        if (args.pixuv == 1) { w.put_var("Upix", resVar.uPix); w.put_var("Vpix", resVar.vPix); }
        if (args.dointerp == 1 && args.putinterp == 1) w.put_var("Occlusion", resVar.occlusion);
        if (args.outrad) {
            const size_t wn = (size_t)nav.nx * nav.ny;
            if (args.putinterp == 0) {
                w.put_var("Rad", resVar.data.data);                                       // ref fw:525-545
                if (args.doc2 == 1) w.put_var("Rad2", resVar.data.data + wn);
                if (args.doc3 == 1) w.put_var("Rad3", resVar.data.data + 2 * wn);
            } else {
                w.put_var("Rad", resVar.dataSValfloat);
                if (args.doc2 == 1) w.put_var("Rad2", resVar.dataSValfloat2);
                if (args.doc3 == 1) w.put_var("Rad3", resVar.dataSValfloat3);
            }
        }
        const int g = (int)nav.gipVal;
        w.put_var(gip, &g);
        w.close();
        return 0;
    } catch (const nc4lite::Error &e) {
        std::cout << e.what() << "\nPOLARWRITE failure\n";
        return NC_ERR;
    }
}

// ref fw:565-700.  U / V are doubles holding the navigated winds x 100 (the shorts uVal / vVal) with scale_factor 0.01.
int oct_mercwrite(std::string fpath, GOESVar &resVar, OFFlags args)
{
    using nc4lite::Type;
    try {
        nc4lite::Writer w(fpath);
        const GOESNAVVar &nav = resVar.nav;
        put_axes_and_time(w, resVar, args, false);
        const std::vector<std::string> yx = {"y", "x"};
        const std::string gm = "Mercator Sphere";
        for (const char *n : {"U", "V"}) {
            w.def_var(n, Type::Double, yx);
            w.put_att(n, "long_name", std::string(n));
            w.put_att(n, "grid_mapping", gm);
            w.put_att(n, "scale_factor", 0.01f);
        }
        w.put_att("U", "units", std::string(args.pixuv == 0 ? "meters per second" : "x-pixels"));
        w.put_att("V", "units", std::string(args.pixuv == 1 ? "y-pixels" : "meters per second"));
        if (args.pixuv == 1) {
            for (const char *n : {"Upix", "Vpix"}) {
                w.def_var(n, Type::Float, yx);
                if (args.dosrsal == 1) w.put_att(n, "long_name", std::string(n));
            }
        }
        if (args.outrad) {
            w.def_var("Rad", Type::Float, yx);
            w.put_att("Rad", "long_name", std::string("Rad"));
            w.put_att("Rad", "grid_mapping", gm);
        }
        const char *gip = "merc_imager_projection";
        w.def_var(gip, Type::Int);
        w.put_att(gip, "long_name", std::string("Mercator_Grid"));
        w.put_att(gip, "grid_mapping_name", std::string("Mercator"));
        w.put_att(gip, "lon1", (double)nav.lon1);
        w.put_att(gip, "R", (double)nav.R);
        put_settings(w, resVar, args, false);
        {
            const size_t wn = (size_t)nav.nx * nav.ny;
            std::vector<double> d(wn);
            for (size_t i = 0; i < wn; i++) d[i] = resVar.uVal[i];
            w.put_var("U", d.data());
            for (size_t i = 0; i < wn; i++) d[i] = resVar.vVal[i];
            w.put_var("V", d.data());
        }
        if (args.pixuv == 1) { w.put_var("Upix", resVar.uPix); w.put_var("Vpix", resVar.vPix); }
        if (args.outrad) w.put_var("Rad", resVar.data.data);
        const int g = (int)nav.gipVal;
        w.put_var(gip, &g);
        w.close();
        return 0;
    } catch (const nc4lite::Error &e) {
        std::cout << e.what() << "\nMERCWRITE failure\n";
        return NC_ERR;
    }
}

int oct_filewrite(std::string fpath, std::string ftype, GOESVar &resVar, OFFlags args)
{
    if (ftype == "GOES") return oct_goeswrite(fpath, resVar, args);
    if (ftype == "POLAR") return oct_polarwrite(fpath, resVar, args);
    if (ftype == "MERC") return oct_mercwrite(fpath, resVar, args);
    std::cout << "File type " << ftype << " is not supported by this build\n";
    return NC_ERR;
}
