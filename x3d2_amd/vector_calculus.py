"""vector_calculus_t mirror (/root/reference/src/vector_calculus.f90):
curl, divergence_v2c, gradient_c2v as sequences of backend operations and
block get/release, in the reference's order."""
from .common import (DIR_X, DIR_Y, DIR_Z, RDR_X2Y, RDR_X2Z, RDR_Y2X, RDR_Y2Z, RDR_Z2X, RDR_Z2Y, X3dError)


class VectorCalculus:
    def __init__(self, backend):
        self.backend = backend

    def curl(self, o_i_hat, o_j_hat, o_k_hat, u, v, w, x_der1st, y_der1st, z_der1st):
        """:40-140"""
        b, al = self.backend, self.backend.allocator
        if any(f.dir != DIR_X for f in (o_i_hat, o_j_hat, o_k_hat, u, v, w)):
            raise X3dError("Error in curl input/output field %dirs: outputs and inputs must be in DIR_X layout.")
        w_y, dwdy_y = al.get_block(DIR_Y), al.get_block(DIR_Y)
        b.reorder(w_y, w, RDR_X2Y)
        b.tds_solve(dwdy_y, w_y, y_der1st)
        b.reorder(o_i_hat, dwdy_y, RDR_Y2X)
        al.release_block(w_y); al.release_block(dwdy_y)
        v_z, dvdz_z = al.get_block(DIR_Z), al.get_block(DIR_Z)
        b.reorder(v_z, v, RDR_X2Z)
        b.tds_solve(dvdz_z, v_z, z_der1st)
        dvdz_x = al.get_block(DIR_X)
        b.reorder(dvdz_x, dvdz_z, RDR_Z2X)
        al.release_block(v_z); al.release_block(dvdz_z)
        b.vecadd(-1.0, dvdz_x, 1.0, o_i_hat)
        al.release_block(dvdz_x)
        u_z, dudz_z = al.get_block(DIR_Z), al.get_block(DIR_Z)
        b.reorder(u_z, u, RDR_X2Z)
        b.tds_solve(dudz_z, u_z, z_der1st)
        dudz_x = al.get_block(DIR_X)
        b.reorder(dudz_x, dudz_z, RDR_Z2X)
        al.release_block(u_z); al.release_block(dudz_z)
        b.tds_solve(o_j_hat, w, x_der1st)
        b.vecadd(1.0, dudz_x, -1.0, o_j_hat)
        al.release_block(dudz_x)
        b.tds_solve(o_k_hat, v, x_der1st)
        u_y, dudy_y = al.get_block(DIR_Y), al.get_block(DIR_Y)
        b.reorder(u_y, u, RDR_X2Y)
        b.tds_solve(dudy_y, u_y, y_der1st)
        dudy_x = al.get_block(DIR_X)
        b.reorder(dudy_x, dudy_y, RDR_Y2X)
        al.release_block(u_y); al.release_block(dudy_y)
        b.vecadd(-1.0, dudy_x, 1.0, o_k_hat)
        al.release_block(dudy_x)

    def divergence_v2c(self, div_u, u, v, w, x_stagder_v2c, x_interpl_v2c, y_stagder_v2c, y_interpl_v2c,
                       z_stagder_v2c, z_interpl_v2c):
        """:142-246"""
        b, al = self.backend, self.backend.allocator
        if div_u.dir != DIR_Z or any(f.dir != DIR_X for f in (u, v, w)):
            raise X3dError("Error in divergence_v2c input/output field dirs: "
                           "output must be in DIR_Z, inputs must be in DIR_X layout.")
        du_x, dv_x, dw_x = (al.get_block(DIR_X) for _ in range(3))
        b.tds_solve(du_x, u, x_stagder_v2c)
        b.tds_solve(dv_x, v, x_interpl_v2c)
        b.tds_solve(dw_x, w, x_interpl_v2c)
        u_y, v_y, w_y = (al.get_block(DIR_Y) for _ in range(3))
        b.reorder(u_y, du_x, RDR_X2Y)
        b.reorder(v_y, dv_x, RDR_X2Y)
        b.reorder(w_y, dw_x, RDR_X2Y)
        for f in (du_x, dv_x, dw_x):
            al.release_block(f)
        du_y, dv_y, dw_y = (al.get_block(DIR_Y) for _ in range(3))
        b.tds_solve(du_y, u_y, y_interpl_v2c)
        b.tds_solve(dv_y, v_y, y_stagder_v2c)
        b.tds_solve(dw_y, w_y, y_interpl_v2c)
        for f in (u_y, v_y, w_y):
            al.release_block(f)
        u_z, w_z = al.get_block(DIR_Z), al.get_block(DIR_Z)
        b.vecadd(1.0, dv_y, 1.0, du_y)
        b.reorder(u_z, du_y, RDR_Y2Z)
        b.reorder(w_z, dw_y, RDR_Y2Z)
        for f in (du_y, dv_y, dw_y):
            al.release_block(f)
        dw_z = al.get_block(DIR_Z)
        b.tds_solve(div_u, u_z, z_interpl_v2c)
        b.tds_solve(dw_z, w_z, z_stagder_v2c)
        b.vecadd(1.0, dw_z, 1.0, div_u)
        for f in (u_z, w_z, dw_z):
            al.release_block(f)

    def gradient_c2v(self, dpdx, dpdy, dpdz, p, x_stagder_c2v, x_interpl_c2v, y_stagder_c2v, y_interpl_c2v,
                     z_stagder_c2v, z_interpl_c2v):
        """:248-332"""
        b, al = self.backend, self.backend.allocator
        if any(f.dir != DIR_X for f in (dpdx, dpdy, dpdz)) or p.dir != DIR_Z:
            raise X3dError("Error in gradient_c2v input/output field dirs: "
                           "outputs must be in DIR_X, input must be in DIR_Z layout.")
        p_sxy_z, dpdz_sxy_z = al.get_block(DIR_Z), al.get_block(DIR_Z)
        b.tds_solve(p_sxy_z, p, z_interpl_c2v)
        b.tds_solve(dpdz_sxy_z, p, z_stagder_c2v)
        p_sxy_y, dpdz_sxy_y = al.get_block(DIR_Y), al.get_block(DIR_Y)
        b.reorder(p_sxy_y, p_sxy_z, RDR_Z2Y)
        b.reorder(dpdz_sxy_y, dpdz_sxy_z, RDR_Z2Y)
        al.release_block(p_sxy_z); al.release_block(dpdz_sxy_z)
        p_sx_y, dpdy_sx_y = al.get_block(DIR_Y), al.get_block(DIR_Y)
        b.tds_solve(p_sx_y, p_sxy_y, y_interpl_c2v)
        b.tds_solve(dpdy_sx_y, p_sxy_y, y_stagder_c2v)
        al.release_block(p_sxy_y)
        dpdz_sx_y = al.get_block(DIR_Y)
        b.tds_solve(dpdz_sx_y, dpdz_sxy_y, y_interpl_c2v)
        al.release_block(dpdz_sxy_y)
        p_sx_x = al.get_block(DIR_X)
        b.reorder(p_sx_x, p_sx_y, RDR_Y2X)
        al.release_block(p_sx_y)
        dpdy_sx_x = al.get_block(DIR_X)
        b.reorder(dpdy_sx_x, dpdy_sx_y, RDR_Y2X)
        al.release_block(dpdy_sx_y)
        dpdz_sx_x = al.get_block(DIR_X)
        b.reorder(dpdz_sx_x, dpdz_sx_y, RDR_Y2X)
        al.release_block(dpdz_sx_y)
        b.tds_solve(dpdx, p_sx_x, x_stagder_c2v)
        b.tds_solve(dpdy, dpdy_sx_x, x_interpl_c2v)
        b.tds_solve(dpdz, dpdz_sx_x, x_interpl_c2v)
        for f in (p_sx_x, dpdy_sx_x, dpdz_sx_x):
            al.release_block(f)
