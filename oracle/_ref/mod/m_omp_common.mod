﻿!mod$ v1 sum:5aa2ec0b70f94be2
module m_omp_common
integer(4),parameter::sz=16_4
end
