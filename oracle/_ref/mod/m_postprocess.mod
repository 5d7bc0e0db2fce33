﻿!mod$ v1 sum:e03d3be8503abc60
!need$ 85f841a7a38b0974 n m_solver
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
module m_postprocess
use m_common,only:dp
use m_common,only:dir_x
use m_common,only:dir_y
use m_common,only:dir_z
use m_common,only:vert
use m_common,only:rdr_x2y
use m_common,only:rdr_x2z
use m_common,only:rdr_y2x
use m_common,only:rdr_z2x
use m_field,only:field_t
use m_solver,only:solver_t
private::dp
private::dir_x
private::dir_y
private::dir_z
private::vert
private::rdr_x2y
private::rdr_x2z
private::rdr_y2x
private::rdr_z2x
private::field_t
private::solver_t
contains
subroutine compute_derived_fields(solver,output_vorticity,output_qcriterion)
class(solver_t),intent(inout)::solver
logical(4),intent(in)::output_vorticity
logical(4),intent(in)::output_qcriterion
end
subroutine compute_pressure_vert(solver)
class(solver_t),intent(inout)::solver
end
end
