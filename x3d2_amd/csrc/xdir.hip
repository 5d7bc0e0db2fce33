// x-direction operators.  In the Cartesian-pitched block an x pencil is
// contiguous, so "one pencil per lane" has lanes 4 KB apart: the generic
// kernels of tds.hip are correct here but uncoalesced.  x3d_xdir_* is the seam
// where the LDS-tile-transposed variant plugs in (kernel K3).
#include "common.h"

int x3d_generic_tds_local(x3d_backend *b, double *du, const double *u, const x3d_tdsops *t, int dir);
int x3d_generic_transeq_local(x3d_backend *b, int dir, double *rhs, const double *u, const double *conv,
                              double nu, const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3);

int x3d_xdir_tds(x3d_backend *b, double *du, const double *u, const x3d_tdsops *t)
{
    return x3d_generic_tds_local(b, du, u, t, X3D_DIR_X);
}

int x3d_xdir_transeq(x3d_backend *b, double *rhs, const double *u, const double *conv, double nu,
                     const x3d_tdsops *t1, const x3d_tdsops *t2, const x3d_tdsops *t3)
{
    return x3d_generic_transeq_local(b, X3D_DIR_X, rhs, u, conv, nu, t1, t2, t3);
}
