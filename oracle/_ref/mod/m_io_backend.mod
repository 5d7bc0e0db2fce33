﻿!mod$ v1 sum:fed7a848d7ae1b45
!need$ f1de5abe9bfe2168 i iso_fortran_env
!need$ 3ca4be32f1385c79 n m_io_base
!need$ f74ae58d325d162e n m_common
module m_io_backend
use,intrinsic::iso_fortran_env,only:stderr=>error_unit
use m_io_base,only:io_reader_t
use m_io_base,only:io_writer_t
use m_io_base,only:io_file_t
use m_io_base,only:io_mode_read
use m_io_base,only:io_mode_write
use m_common,only:dp
use m_common,only:i8
private::stderr
private::io_reader_t
private::io_writer_t
private::io_file_t
private::io_mode_read
private::io_mode_write
private::dp
private::i8
logical(4),private,save::write_warning_shown
integer(4),parameter::io_backend_dummy=0_4
integer(4),parameter::io_backend_adios2=1_4
type,private,extends(io_file_t)::io_dummy_file_t
logical(4)::is_open=.false._4
contains
procedure::close=>file_close_dummy
procedure::begin_step=>file_begin_step_dummy
procedure::end_step=>file_end_step_dummy
procedure::is_file_functional=>is_file_functional_dummy
end type
type,private,extends(io_reader_t)::io_dummy_reader_t
logical(4)::initialised=.false._4
contains
procedure::init=>reader_init_dummy
procedure::open=>reader_open_dummy
procedure::finalise=>reader_finalise_dummy
procedure::read_data_i8=>read_data_i8_dummy
procedure::read_data_integer=>read_data_integer_dummy
procedure::read_data_real=>read_data_real_dummy
procedure::read_data_array_3d=>read_data_array_3d_dummy
end type
type,private,extends(io_writer_t)::io_dummy_writer_t
logical(4)::initialised=.false._4
contains
procedure::init=>writer_init_dummy
procedure::open=>writer_open_dummy
procedure::finalise=>writer_finalise_dummy
procedure::write_data_i8=>write_data_i8_dummy
procedure::write_data_integer=>write_data_integer_dummy
procedure::write_data_real=>write_data_real_dummy
procedure::write_data_array_3d=>write_data_array_3d_dummy
procedure::write_attribute_string=>write_attribute_string_dummy
procedure::write_attribute_array_1d_real=>write_attribute_array_1d_real_dummy
end type
private::report_read_error
private::file_close_dummy
private::file_begin_step_dummy
private::file_end_step_dummy
private::reader_init_dummy
private::reader_open_dummy
private::is_file_functional_dummy
private::read_data_i8_dummy
private::read_data_integer_dummy
private::read_data_real_dummy
private::read_data_array_3d_dummy
private::reader_finalise_dummy
private::writer_init_dummy
private::writer_open_dummy
private::write_data_i8_dummy
private::write_data_integer_dummy
private::write_data_real_dummy
private::write_data_array_3d_dummy
private::writer_finalise_dummy
private::write_attribute_string_dummy
private::write_attribute_array_1d_real_dummy
contains
subroutine allocate_io_reader(reader)
class(io_reader_t),allocatable,intent(out)::reader
end
subroutine allocate_io_writer(writer)
class(io_writer_t),allocatable,intent(out)::writer
end
function get_default_backend() result(backend)
integer(4)::backend
end
subroutine report_read_error(variable_name)
character(*,1),intent(in)::variable_name
end
subroutine file_close_dummy(self)
class(io_dummy_file_t),intent(inout)::self
end
subroutine file_begin_step_dummy(self)
class(io_dummy_file_t),intent(inout)::self
end
subroutine file_end_step_dummy(self)
class(io_dummy_file_t),intent(inout)::self
end
subroutine reader_init_dummy(self,comm,name)
class(io_dummy_reader_t),intent(inout)::self
integer(4),intent(in)::comm
character(*,1),intent(in)::name
end
function reader_open_dummy(self,filename,mode,comm) result(file_handle)
class(io_dummy_reader_t),intent(inout)::self
character(*,1),intent(in)::filename
integer(4),intent(in)::mode
integer(4),intent(in)::comm
class(io_file_t),allocatable::file_handle
end
function is_file_functional_dummy(self) result(is_functional)
class(io_dummy_file_t),intent(in)::self
logical(4)::is_functional
end
subroutine read_data_i8_dummy(self,variable_name,value,file_handle)
class(io_dummy_reader_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(8),intent(out)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine read_data_integer_dummy(self,variable_name,value,file_handle)
class(io_dummy_reader_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(4),intent(out)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine read_data_real_dummy(self,variable_name,value,file_handle)
class(io_dummy_reader_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(out)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine read_data_array_3d_dummy(self,variable_name,array,file_handle,shape_dims,start_dims,count_dims)
class(io_dummy_reader_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(inout)::array(:,:,:)
class(io_file_t),intent(inout)::file_handle
integer(8),intent(in),optional::shape_dims(1_8:3_8)
integer(8),intent(in),optional::start_dims(1_8:3_8)
integer(8),intent(in),optional::count_dims(1_8:3_8)
end
subroutine reader_finalise_dummy(self)
class(io_dummy_reader_t),intent(inout)::self
end
subroutine writer_init_dummy(self,comm,name)
class(io_dummy_writer_t),intent(inout)::self
integer(4),intent(in)::comm
character(*,1),intent(in)::name
end
function writer_open_dummy(self,filename,mode,comm) result(file_handle)
class(io_dummy_writer_t),intent(inout)::self
character(*,1),intent(in)::filename
integer(4),intent(in)::mode
integer(4),intent(in)::comm
class(io_file_t),allocatable::file_handle
end
subroutine write_data_i8_dummy(self,variable_name,value,file_handle)
class(io_dummy_writer_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(8),intent(in)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine write_data_integer_dummy(self,variable_name,value,file_handle)
class(io_dummy_writer_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(4),intent(in)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine write_data_real_dummy(self,variable_name,value,file_handle,use_sp)
class(io_dummy_writer_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(in)::value
class(io_file_t),intent(inout)::file_handle
logical(4),intent(in),optional::use_sp
end
subroutine write_data_array_3d_dummy(self,variable_name,array,file_handle,shape_dims,start_dims,count_dims,use_sp)
class(io_dummy_writer_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(in)::array(:,:,:)
class(io_file_t),intent(inout)::file_handle
integer(8),intent(in)::shape_dims(1_8:3_8)
integer(8),intent(in)::start_dims(1_8:3_8)
integer(8),intent(in)::count_dims(1_8:3_8)
logical(4),intent(in),optional::use_sp
end
subroutine writer_finalise_dummy(self)
class(io_dummy_writer_t),intent(inout)::self
end
subroutine write_attribute_string_dummy(self,attribute_name,value,file_handle)
class(io_dummy_writer_t),intent(inout)::self
character(*,1),intent(in)::attribute_name
character(*,1),intent(in)::value
class(io_file_t),intent(inout)::file_handle
end
subroutine write_attribute_array_1d_real_dummy(self,attribute_name,values,file_handle)
class(io_dummy_writer_t),intent(inout)::self
character(*,1),intent(in)::attribute_name
real(8),intent(in)::values(:)
class(io_file_t),intent(inout)::file_handle
end
end
