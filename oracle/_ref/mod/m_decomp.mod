﻿!mod$ v1 sum:3ad2cd360e651cee
module m_decomp
contains
function is_avail_2decomp() result(avail)
logical(4)::avail
end
subroutine decomposition_2decomp(grid,par)
use m_mesh_content,only:grid_t
use m_mesh_content,only:par_t
class(grid_t),intent(inout)::grid
class(par_t),intent(inout)::par
end
end
