/* vof_oracle.h -- CPU oracle (test infrastructure only; see vof_oracle.c header).
 * "parity unpinned" for the whole solver; partial pins listed in vof_oracle.c. */
#ifndef OCT_VOF_ORACLE_H
#define OCT_VOF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oct_oracle_params {
    double alpha, lambda, lambdac, scaleF;  /* OFFlags fields read at ref .cu:1229-1241 */
    int kiters, liters, cgiters, dozim;
} oct_oracle_params;

/* Optional per-stage dump hook: data is nplanes planes of nx*ny floats. */
typedef void (*oct_oracle_trace_fn)(void *user, const char *tag, int level, int gnc, int l,
                                    const float *data, int nx, int ny, int nplanes);
typedef struct oct_oracle_trace { oct_oracle_trace_fn cb; void *user; } oct_oracle_trace;

typedef struct oct_oracle_level {
    int xi, yi, nc;
    const float *img1, *img2, *gx1, *gy1, *gx2, *gy2, *gxx, *gxy, *gyy;
} oct_oracle_level;

typedef struct oct_oracle_system {  /* CSR as the reference lays it out */
    int nrows; long nnz;
    float *val; int *row; int *col; int *rowptr; float *diag; float *rhs;
} oct_oracle_system;

typedef struct oct_oracle_planes { float *a1, *a2, *a4, *a5, *a6, *a7, *a8, *bu, *bv; } oct_oracle_planes;
typedef struct oct_oracle_cgwork { float *z, *p, *rk, *tmp; int *ident; } oct_oracle_cgwork;

int   oct_oracle_num_threads(void);
void  oct_oracle_set_threads(int n);
void  oct_oracle_level_dims(int nx, int ny, float factor, int *lx, int *ly);
float oct_oracle_level_factor(float scale, int kiters, int k);
int   oct_oracle_blur_halfwidth(float factor);
void  oct_oracle_gauss_taps(float factor, int fs, float *gk);
void  oct_oracle_blur_rows(const float *in, float *out, const float *gk, int nx, int ny, int nc, int fs);
void  oct_oracle_blur_cols(const float *in, float *out, const float *gk, int nx, int ny, int nc, int fs);
float oct_oracle_bicubic(const float *src, float uu, float vv, int nx, int ny);
void  oct_oracle_decimate(const float *blurred, float *out, int nx, int ny, int nc, float factor);
void  oct_oracle_gradient(const float *f, float *gx, float *gy, int xi, int yi, int nc);
void  oct_oracle_upsample_flow(const float *coarse, float *fine, int nx, int ny, int nxx, int nyy, float sf);
float oct_oracle_clamp_coord(float x, int n, int *hit);                       /* ref .cu:26-41 */
float oct_oracle_bilinear(float px, float py, int xi, int yi, float f11, float f21, float f12, float f22,
                          float *p4, int *cell_xy, int *hit_xy);             /* ref .cu:56-71, 727-747 */
long  oct_oracle_nnz_before(long n, int ii, int jj, int xi, int yi);
void  oct_oracle_assemble(const oct_oracle_level *L, const float *u, const float *v,
                          const float *ut, const float *vt, double al1, double alpha,
                          double lambda_over_alpha, float lambdac, int dozim,
                          oct_oracle_system *S, oct_oracle_planes *P);
void  oct_oracle_spmv(const float *val, const int *rowptr, const int *col, const float *x,
                      long nnz, int nrows, float *y);
/* 0 = one-thread running sum (default), >0 = the reference's grid schedule with that many threads */
void  oct_oracle_set_dot_schedule(int threads);
int   oct_oracle_pcg(oct_oracle_system *S, float *x, float tol, int maxit, oct_oracle_cgwork *W);

/* Whole solve.  u/v are in-out (first guess in, flow out).  Returns the number
 * of PCG iterations executed (>=0) or a negative error. */
int   oct_oracle_vof(const float *img1, const float *img2, int nx, int ny, int nc,
                     float *u, float *v, const oct_oracle_params *prm, const oct_oracle_trace *tr);

/* ---- pix2uv (pix2uv_oracle.c) ---- */
typedef struct oct_oracle_nav {   /* the GOESNAVVar fields ref oct_pix2uv_cuda.cu reads */
    double pph, req, rpol, lam0;
    float xScale, xOffset, yScale, yOffset, g2xOffset, g2yOffset;
    float lat1, lon1, lon0, R;
    int minX, minY;
    int nx, ny;
} oct_oracle_nav;

/* mode: 0 geostationary fixed grid, 1 polar (-Polar), 2 mercator (-Merc).
 * Returns 0, or 1 when the sector-moved guard zeroed everything. */
/* which multiply-add sites of the navigation are fused (bit mask, pix2uv_oracle.c; 0 = none: the strict two-rounding forms) */
void oct_oracle_pix2uv_fma_sites(unsigned mask);
int oct_oracle_pix2uv_nsites(void);
int oct_oracle_pix2uv(const oct_oracle_nav *nav, double t1, double t2, const float *u, const float *v,
                      int pixuv, int mode, short *ur, short *vr, short *ur2, short *vr2, float *dT);

/* ---- navcal (navcal_oracle.c) ---- */
typedef struct oct_oracle_navcal_params {
    float xScale, xOffset, yScale, yOffset, radScale, radOffset;
    float rpol, req, H, lam0;
    float fk1, fk2, bc1, bc2, kap1;
    float maxin, minin, maxout, minout;
    int cal, donav;
    int minx, maxx, miny, maxy;
} oct_oracle_navcal_params;
void oct_oracle_navcal(const short *data2, const short *x, const short *y, int nx, int ny,
                       const oct_oracle_navcal_params *p, float *data3, float *lat, float *lon,
                       short *data2s, short *xs, short *ys);

typedef struct oct_oracle_proj_navcal_params {
    float xScale, xOffset, yScale, yOffset, lon0, lat1, R;   /* lon0, lat1 in degrees */
    int donav, mode;                                          /* mode 1 polar, 2 mercator */
    int minx, maxx, miny, maxy;
} oct_oracle_proj_navcal_params;
void oct_oracle_proj_navcal(const float *data2, const short *x, const short *y, int nx, int ny,
                            const oct_oracle_proj_navcal_params *p, float *data3, float *lat, float *lon,
                            short *data2s, short *xs, short *ys);

/* ---- uv2pix / srsal (post_oracle.c) ---- */
void oct_oracle_uv2pix(const oct_oracle_nav *nav, double t1, double t2, float *u_inout, float *v_inout,
                       const float *lat, const float *lon, const short *gx, const short *gy);
void oct_oracle_srsal(float *u, float *v, const float *cth, int nx, int ny, float *uo, float *vo);

/* ---- patch matching, -sosm (sosm_oracle.c) ---- */
int  oct_oracle_sosm_spiral(int srad, int *nm);
void oct_oracle_sosm(const float *img1, const float *img2, float *u_inout, float *v_inout, int nx, int ny, int rad, int srad);

#ifdef __cplusplus
}
#endif
#endif
