﻿!mod$ v1 sum:fadd42cafe0c8e6b
!need$ fed7a848d7ae1b45 n m_io_backend
!need$ 3ca4be32f1385c79 n m_io_base
!need$ f74ae58d325d162e n m_common
module m_io_session
use m_common,only:dp
use m_common,only:i8
use m_io_base,only:io_reader_t
use m_io_base,only:io_writer_t
use m_io_base,only:io_file_t
use m_io_base,only:io_mode_read
use m_io_base,only:io_mode_write
use m_io_backend,only:allocate_io_reader
use m_io_backend,only:allocate_io_writer
private::dp
private::i8
private::io_reader_t
private::io_writer_t
private::io_file_t
private::io_mode_read
private::io_mode_write
private::allocate_io_reader
private::allocate_io_writer
type,private::io_session_base_t
class(io_file_t),allocatable,private::file
logical(4),private::is_open=.false._4
logical(4),private::is_functional=.true._4
contains
procedure::is_session_open
procedure::is_session_functional
procedure::close=>session_base_close
end type
type,extends(io_session_base_t)::reader_session_t
class(io_reader_t),allocatable,private::reader
contains
procedure::open=>reader_session_open
generic::read_data=>read_data_i8
generic::read_data=>read_data_integer
generic::read_data=>read_data_real
generic::read_data=>read_data_array_3d
procedure,private::read_data_i8
procedure,private::read_data_integer
procedure,private::read_data_real
procedure,private::read_data_array_3d
final::reader_session_finaliser
end type
type,extends(io_session_base_t)::writer_session_t
class(io_writer_t),allocatable,private::writer
contains
procedure::open=>writer_session_open
procedure::begin_step=>writer_session_begin_step
procedure::end_step=>writer_session_end_step
generic::write_data=>write_data_i8
generic::write_data=>write_data_integer
generic::write_data=>write_data_real
generic::write_data=>write_data_array_3d
procedure,private::write_data_i8
procedure,private::write_data_integer
procedure,private::write_data_real
procedure,private::write_data_array_3d
procedure::write_attribute=>session_write_attribute
final::writer_session_finaliser
end type
private::is_session_open
private::is_session_functional
private::session_base_close
private::reader_session_open
private::read_data_i8
private::read_data_integer
private::read_data_real
private::read_data_array_3d
private::writer_session_open
private::write_data_i8
private::write_data_integer
private::write_data_real
private::write_data_array_3d
private::session_write_attribute
private::writer_session_begin_step
private::writer_session_end_step
private::reader_session_finaliser
private::writer_session_finaliser
contains
function is_session_open(self)
class(io_session_base_t),intent(in)::self
logical(4)::is_session_open
end
function is_session_functional(self)
class(io_session_base_t),intent(in)::self
logical(4)::is_session_functional
end
subroutine session_base_close(self)
class(io_session_base_t),intent(inout)::self
end
subroutine reader_session_open(self,filename,comm)
class(reader_session_t),intent(inout)::self
character(*,1),intent(in)::filename
integer(4),intent(in)::comm
end
subroutine read_data_i8(self,variable_name,value)
class(reader_session_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(8),intent(out)::value
end
subroutine read_data_integer(self,variable_name,value)
class(reader_session_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(4),intent(out)::value
end
subroutine read_data_real(self,variable_name,value)
class(reader_session_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(out)::value
end
subroutine read_data_array_3d(self,variable_name,array,start_dims,count_dims,shape_dims)
class(reader_session_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(inout)::array(:,:,:)
integer(8),intent(in),optional::start_dims(1_8:3_8)
integer(8),intent(in),optional::count_dims(1_8:3_8)
integer(8),intent(in),optional::shape_dims(1_8:3_8)
end
subroutine writer_session_open(self,filename,comm)
class(writer_session_t),intent(inout)::self
character(*,1),intent(in)::filename
integer(4),intent(in)::comm
end
subroutine write_data_i8(self,variable_name,value)
class(writer_session_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(8),intent(in)::value
end
subroutine write_data_integer(self,variable_name,value)
class(writer_session_t),intent(inout)::self
character(*,1),intent(in)::variable_name
integer(4),intent(in)::value
end
subroutine write_data_real(self,variable_name,value,use_sp)
class(writer_session_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(in)::value
logical(4),intent(in),optional::use_sp
end
subroutine write_data_array_3d(self,variable_name,array,shape_dims,start_dims,count_dims,use_sp)
class(writer_session_t),intent(inout)::self
character(*,1),intent(in)::variable_name
real(8),intent(in)::array(:,:,:)
integer(8),intent(in)::shape_dims(1_8:3_8)
integer(8),intent(in)::start_dims(1_8:3_8)
integer(8),intent(in)::count_dims(1_8:3_8)
logical(4),intent(in),optional::use_sp
end
subroutine session_write_attribute(self,attribute_name,attribute_value)
class(writer_session_t),intent(inout)::self
character(*,1),intent(in)::attribute_name
character(*,1),intent(in)::attribute_value
end
subroutine writer_session_begin_step(self)
class(writer_session_t),intent(inout)::self
end
subroutine writer_session_end_step(self)
class(writer_session_t),intent(inout)::self
end
subroutine reader_session_finaliser(self)
type(reader_session_t)::self
end
subroutine writer_session_finaliser(self)
type(writer_session_t)::self
end
end
