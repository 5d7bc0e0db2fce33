"""Constants mirrored from the reference's m_common
(/root/reference/src/common.f90:23-39, 84-88)."""

DIR_X, DIR_Y, DIR_Z, DIR_C = 1, 2, 3, 4
RDR_X2Y, RDR_X2Z, RDR_Y2X, RDR_Y2Z, RDR_Z2X, RDR_Z2Y = 12, 13, 21, 23, 31, 32
RDR_C2X, RDR_C2Y, RDR_C2Z, RDR_X2C, RDR_Y2C, RDR_Z2C = 41, 42, 43, 14, 24, 34
VERT, CELL, NULL_LOC = 0, 1110, -1
X_FACE, Y_FACE, Z_FACE = 1100, 1010, 110
X_EDGE, Y_EDGE, Z_EDGE = 10, 100, 1000
BC_PERIODIC, BC_NEUMANN, BC_DIRICHLET, BC_HALO = 0, 1, 2, -1
N_HALO = 4
BC_NAMES = {"periodic": BC_PERIODIC, "neumann": BC_NEUMANN, "dirichlet": BC_DIRICHLET}


def move_data_loc(in_data_loc, direction, move):
    """src/common.f90:84-88"""
    return in_data_loc + move * 10 ** direction


def get_rdr_from_dirs(dir_from, dir_to):
    """src/common.f90 get_rdr_from_dirs: 0 when no reorder is needed"""
    return 0 if dir_from == dir_to else 10 * dir_from + dir_to


class X3dError(RuntimeError):
    """the reference `error stop`s; the mirror raises"""
