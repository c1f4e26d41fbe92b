// octane_types.hpp -- the host-side boundary types of OCTANE's flow path, layout-compatible with the
// reference's include/image.h, include/offlags.h and include/goesread.h (same class names, same members in
// the same order with the same types), so that host code written against the reference compiles and links
// against this library unchanged.  tests/test_host_abi.py checks sizeof/offsetof of every member against the
// reference's own headers where those are present.  include/image.h, offlags.h and goesread.h forward here.
#pragma once
#include <cstdlib>
#include <string>

// ---- ref include/image.h:3-24 ------------------------------------------------------------------------------
// Non-owning view of a channel-planar float image: data[i + nrow*j + nrow*ncol*c].  NOTE the reference's
// naming: nrow is the x extent (fastest index) and ncol the y extent (ref src/oct_fileread.cc:281).
class Image {
  public:
    float *data;
    int nrow, ncol, nchannels;

    Image() : nrow(0), ncol(0), nchannels(0) {}
    Image(int x, int y, int c) : nrow(x), ncol(y), nchannels(c) {}
    void setdims(int x, int y, int c)
    {
        nrow = x;
        ncol = y;
        nchannels = c;
    }
};

// ---- ref include/offlags.h:4-72 ----------------------------------------------------------------------------
// Every command-line switch of `octane` (ref src/main.cc:42-50, defaults :53-108).  The variational solver reads
// alpha, lambda, lambdac, kiters, liters, cgiters, dozim, scsig, scaleF and setdevice (ref .cu:1229-1254);
// pix2uv reads pixuv, dopolar, domerc and setdevice (ref src/oct_pix2uv_cuda.cu:279-297).
class OFFlags {
  public:
    // algorithm / grid selection
    int farn;          // -farn (disabled in the reference)
    int pixuv;         // -pd: output pixel displacements instead of navigated winds
    int dopolar;       // -Polar
    int domerc;        // -Merc
    int doahi;         // -ahi
    int dosrsal;       // -srsal
    int dososm;        // -sosm: patch matching instead of the variational solver
    int dofirstguess;  // -firstguess <file>
    std::string ftype; // "GOES" | "POLAR" | "MERC"
    int dointerp;      // -interp
    int docorn;        // -corn (sets 0 in the reference, src/main.cc:270-273)
    int putinterp;     // internal
    int interpcth;     // -nncth clears it
    int doinv;         // -inv
    int doctt;         // -ctt
    int dozim;         // 1 = Zimmer normalisation, -brox clears it
    int oftype;        // 1 Zimmer, 2 Farneback, 3 Brox, 4 patch match (src/main.cc:369-388)
    int doc2;          // -ic21/-ic22 given
    int doc3;          // -ic31/-ic32 given
    int ir;            // -ir
    int rad;           // -rad
    int srad;          // -srad
    int setdevice;     // -set_device N (stored 0-based)
    // Farneback leftovers
    float fpyr_scale;
    float flevels;
    int fwinsize;
    int fiterations;
    int poly_n;
    float poly_sigma;
    float deltat;      // -deltat
    int uif;
    int fg;
    int doCTH;         // -i1cth given
    // variational solver
    double lambda;     // -lambda
    double alpha;      // -alpha
    double alpha2;     // -alpha2 (unused)
    double lambdac;    // -lambdac
    double scsig;      // -scsig (stores the SQUARE of its argument, src/main.cc:229)
    double filtsigma;
    double scaleF;     // pyramid scale factor, 0.5, no switch
    int kiters;        // -kiters
    int liters;        // -liters
    int cgiters;       // documented as -cgiters but never parsed (src/main.cc:144 vs :42-50)
    int miters;
    int setnorms;
    float NormMax;     // -normmax
    float NormMin;     // -normmin
    float NormMax2;
    float NormMin2;
    float NormMax3;
    float NormMin3;
    bool outnav;       // -no_outnav clears
    bool outraw;       // -no_outraw clears
    bool outrad;       // -no_outrad clears
    bool outctp;       // -no_outctp clears
    bool setNormMax;
    bool setNormMin;
    bool setNormMax2;
    bool setNormMin2;
    bool setNormMax3;
    bool setNormMin3;
};

// ---- ref include/goesread.h:3-57 ---------------------------------------------------------------------------
// Projection constants of one file.  pix2uv passes this object BY VALUE into its kernel in the reference
// (ref src/oct_pix2uv_cuda.cu:174); here only the fields listed in octane_nav (octane_vof.h) cross the C-ABI.
class GOESNAVVar {
  public:
    double pph, req, rpol, lam0, inverse_flattening, lat0;
    float gipVal, xScale, xOffset, yScale, yOffset, g2xOffset, g2yOffset, fk1, fk2, bc1, bc2, lpo, kap1, radScale, radOffset;
    float fk12, fk22, bc12, bc22, kap12, fk13, fk23, bc13, bc23, kap13;
    float radScale2, radScale3, radOffset2, radOffset3;
    long nx2, ny2, nx3, ny3;
    long nx, ny, CTHx, CTHy;
    int minXc, maxXc, minYc, maxYc, minX, minY, maxX, maxY;
    float lat1, lon1, lon0, R;
};

// Everything read from / written to one GOES-R (or polar / mercator) file.
class GOESVar {
  public:
    float *latVal;
    float *lonVal;
    short *x;
    short *y;
    short *CTP, *CTT;
    unsigned char *CTI;
    float *dataVal, *dataVal2, *dataVal3;
    Image data;                 // the solver's input: 0..255-normalised radiances, channel-planar
    float *dataVal2i, *dataVal3i;
    short *occlusion;
    short *uVal;                // navigated wind x100 (cm/s)        <- pix2uv
    short *vVal;
    short *uVal2;               // raw pixel displacement x100        <- pix2uv
    short *vVal2;
    short *cnrarr;
    float *uPix;                // flow in pixels: first guess in, solver result out
    float *vPix;
    double *u1;
    double *v1;
    float *UFG;
    float *VFG;
    double *u2;
    double *v2;
    short *accel;
    float *CTHVal;
    float *CTTVal;
    unsigned char *CTHInv;
    short *dataSVal, *dataSVal2, *dataSVal3;
    short *dataSValint, *dataSValint2, *dataSValint3;
    float *dataSValfloat, *dataSValfloat2, *dataSValfloat3;
    double t;                   // scan time, seconds
    double tint;
    float dT;                   // t2 - t1, set by pix2uv
    float frdt;
    int band, band2, band3;
    GOESNAVVar nav;
    std::string tUnits;
};
