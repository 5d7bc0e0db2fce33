﻿!mod$ v1 sum:3ca4be32f1385c79
!need$ f74ae58d325d162e n m_common
module m_io_base
use m_common,only:dp
use m_common,only:i8
private::dp
private::i8
integer(4),parameter::io_mode_read=1_4
integer(4),parameter::io_mode_write=2_4
type::io_file_t
contains
procedure::close=>base_close
procedure::begin_step=>base_begin_step
procedure::end_step=>base_end_step
procedure::is_file_functional=>base_is_file_functional
end type
type::io_reader_t
contains
procedure::init=>base_reader_init
procedure::open=>base_reader_open
procedure::finalise=>base_reader_finalise
generic::read_data=>read_data_i8
generic::read_data=>read_data_integer
generic::read_data=>read_data_real
generic::read_data=>read_data_array_3d
procedure::read_data_i8
procedure::read_data_integer
procedure::read_data_real
procedure::read_data_array_3d
end type
type::io_writer_t
contains
procedure::init=>base_writer_init
procedure::open=>base_writer_open
procedure::finalise=>base_writer_finalise
generic::write_data=>write_data_i8
generic::write_data=>write_data_integer
generic::write_data=>write_data_real
generic::write_data=>write_data_array_3d
procedure::write_data_i8
procedure::write_data_integer
procedure::write_data_real
procedure::write_data_array_3d
generic::write_attribute=>write_attribute_string
generic::write_attribute=>write_attribute_array_1d_real
procedure::write_attribute_string
procedure::write_attribute_array_1d_real
end type
private::base_close
private::base_begin_step
private::base_end_step
private::base_reader_init
private::base_reader_open
private::base_reader_finalise
private::base_writer_init
private::base_writer_open
private::base_writer_finalise
private::base_is_file_functional
private::read_data_i8
private::read_data_integer
private::read_data_real
private::read_data_array_3d
private::write_data_i8
private::write_data_integer
private::write_data_real
private::write_data_array_3d
private::write_attribute_string
private::write_attribute_array_1d_real
contains
subroutine base_close(self)
class(io_file_t),intent(inout)::self
end
subroutine base_begin_step(self)
class(io_file_t),intent(inout)::self
end
subroutine base_end_step(self)
class(io_file_t),intent(inout)::self
end
subroutine base_reader_init(self,comm,name)
class(io_reader_t),intent(inout)::self
integer(4),intent(in)::comm
character(*,1),intent(in)::name
end
function base_reader_open(self,filename,mode,comm) result(file_handle)
class(io_reader_t),intent(inout)::self
character(*,1),intent(in)::filename
integer(4),intent(in)::mode
integer(4),intent(in)::comm
class(io_file_t),allocatable::file_handle
end
subroutine base_reader_finalise(self)
class(io_reader_t),intent(inout)::self
end
subroutine base_writer_init(self,comm,name)
class(io_writer_t),intent(inout)::self
integer(4),intent(in)::comm
character(*,1),intent(in)::name
end
function base_writer_open(self,filename,mode,comm) result(file_handle)
class(io_writer_t),intent(inout)::self
character(*,1),intent(in)::filename
integer(4),intent(in)::mode
integer(4),intent(in)::comm
class(io_file_t),allocatable::file_handle
end
subroutine base_writer_finalise(self)
class(io_writer_t),intent(inout)::self
end
function base_is_file_functional(self) result(is_functional)
class(io_file_t),intent(in)::self
logical(4)::is_functional
end
subroutine read_data_i8(self,variable_name,value,file_handle)
class(io_reader_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(8),intent(out)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine read_data_integer(self,variable_name,value,file_handle)
class(io_reader_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(4),intent(out)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine read_data_real(self,variable_name,value,file_handle)
class(io_reader_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(out)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine read_data_array_3d(self,variable_name,array,file_handle,shape_dims,start_dims,count_dims)
class(io_reader_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(inout)::array(:,:,:)
class(io_file_t),intent(inout)::file_handle
integer(8),intent(in),optional::shape_dims(1_8:3_8)
integer(8),intent(in),optional::start_dims(1_8:3_8)
integer(8),intent(in),optional::count_dims(1_8:3_8)
end
subroutine write_data_i8(self,variable_name,value,file_handle)
class(io_writer_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(8),intent(in)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine write_data_integer(self,variable_name,value,file_handle)
class(io_writer_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(4),intent(in)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine write_data_real(self,variable_name,value,file_handle,use_sp)
class(io_writer_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(in)::value
class(io_file_t),intent(inout)::file_handle
logical(4),intent(in),optional::use_sp
end
subroutine write_data_array_3d(self,variable_name,array,file_handle,shape_dims,start_dims,count_dims,use_sp)
class(io_writer_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(in)::array(:,:,:)
class(io_file_t),intent(inout)::file_handle
integer(8),intent(in)::shape_dims(1_8:3_8)
integer(8),intent(in)::start_dims(1_8:3_8)
integer(8),intent(in)::count_dims(1_8:3_8)
logical(4),intent(in),optional::use_sp
end
subroutine write_attribute_string(self,attribute_name,value,file_handle)
class(io_writer_t),intent(inout)::self
character(*,1),intent(in)::attribute_name
character(*,1),intent(in)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine write_attribute_array_1d_real(self,attribute_name,values,file_handle)
class(io_writer_t),intent(inout)::self
character(*,1),intent(in)::attribute_name
real(8),intent(in)::values(:)
class(io_file_t),intent(inout)::file_handle
end
end
