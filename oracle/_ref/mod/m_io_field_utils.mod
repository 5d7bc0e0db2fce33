﻿!mod$ v1 sum:a1f26d8334a87c7c
!need$ 85f841a7a38b0974 n m_solver
!need$ 3ca4be32f1385c79 n m_io_base
!need$ f74ae58d325d162e n m_common
!need$ d9a8bda24462498c n m_field
module m_io_field_utils
use m_common,only:dp
use m_common,only:i8
use m_common,only:dir_c
use m_field,only:field_t
use m_solver,only:solver_t
use m_io_base,only:io_file_t
use m_io_base,only:io_writer_t
private::dp
private::i8
private::dir_c
private::field_t
private::solver_t
private::io_file_t
private::io_writer_t
type::field_buffer_map_t
character(32_4,1)::field_name
real(8),allocatable::buffer(:,:,:)
end type
type::field_ptr_t
class(field_t),pointer::ptr=>NULL()
end type
intrinsic::null
private::null
private::parse_species_snapshot_field
contains
function stride_data(input_data,dims,stride,output_dims_out) result(output_data)
real(8),intent(in)::input_data(:,:,:)
integer(4),intent(in)::dims(1_8:3_8)
integer(4),intent(in)::stride(1_8:3_8)
integer(4),intent(out)::output_dims_out(1_8:3_8)
real(8),allocatable::output_data(:,:,:)
end
subroutine stride_data_to_buffer(input_data,dims,stride,out_buffer,output_dims_out)
real(8),intent(in)::input_data(:,:,:)
integer(4),intent(in)::dims(1_8:3_8)
integer(4),intent(in)::stride(1_8:3_8)
real(8),allocatable,intent(inout)::out_buffer(:,:,:)
integer(4),intent(out)::output_dims_out(1_8:3_8)
end
subroutine get_output_dimensions(shape_dims,start_dims,count_dims,stride_factors,output_shape,output_start,output_count,output_dims_local,last_shape_dims,last_stride_factors,last_output_shape)
integer(8),intent(in)::shape_dims(1_8:3_8)
integer(8),intent(in)::start_dims(1_8:3_8)
integer(8),intent(in)::count_dims(1_8:3_8)
integer(4),intent(in)::stride_factors(1_8:3_8)
integer(8),intent(out)::output_shape(1_8:3_8)
integer(8),intent(out)::output_start(1_8:3_8)
integer(8),intent(out)::output_count(1_8:3_8)
integer(4),intent(out)::output_dims_local(1_8:3_8)
integer(8),intent(inout),optional::last_shape_dims(1_8:3_8)
integer(4),intent(inout),optional::last_stride_factors(1_8:3_8)
integer(8),intent(inout),optional::last_output_shape(1_8:3_8)
end
subroutine generate_coordinates(solver,writer,file,shape_dims,start_dims,count_dims,data_loc,coords_x,coords_y,coords_z)
class(solver_t),intent(in)::solver
class(io_writer_t),intent(inout)::writer
class(io_file_t),intent(inout)::file
integer(8),intent(in)::shape_dims(1_8:3_8)
integer(8),intent(in)::start_dims(1_8:3_8)
integer(8),intent(in)::count_dims(1_8:3_8)
integer(4),intent(in)::data_loc
real(8),allocatable,intent(inout)::coords_x(:,:,:)
real(8),allocatable,intent(inout)::coords_y(:,:,:)
real(8),allocatable,intent(inout)::coords_z(:,:,:)
end
function parse_species_snapshot_field(field_name,species_index)
character(*,1),intent(in)::field_name
integer(4),intent(out)::species_index
logical(4)::parse_species_snapshot_field
end
subroutine setup_field_arrays(solver,field_names,field_ptrs,host_fields)
class(solver_t),intent(in)::solver
character(*,1),intent(in)::field_names(:)
type(field_ptr_t),allocatable,intent(out)::field_ptrs(:)
type(field_ptr_t),allocatable,intent(out)::host_fields(:)
end
subroutine cleanup_field_arrays(solver,field_ptrs,host_fields)
class(solver_t),intent(in)::solver
type(field_ptr_t),allocatable,intent(inout)::field_ptrs(:)
type(field_ptr_t),allocatable,intent(inout)::host_fields(:)
end
subroutine prepare_field_buffers(solver,stride_factors,field_names,data_loc,field_buffers,last_shape_dims,last_stride_factors,last_output_shape)
class(solver_t),intent(in)::solver
integer(4),intent(in)::stride_factors(1_8:3_8)
character(*,1),intent(in)::field_names(:)
integer(4),intent(in)::data_loc
type(field_buffer_map_t),allocatable,intent(inout)::field_buffers(:)
integer(8),intent(inout)::last_shape_dims(1_8:3_8)
integer(4),intent(inout)::last_stride_factors(1_8:3_8)
integer(8),intent(inout)::last_output_shape(1_8:3_8)
end
subroutine write_single_field_to_buffer(field_name,host_field,solver,stride_factors,data_loc,field_buffers,last_shape_dims,last_stride_factors,last_output_shape)
character(*,1),intent(in)::field_name
class(field_t),pointer::host_field
class(solver_t),intent(in)::solver
integer(4),intent(in)::stride_factors(1_8:3_8)
integer(4),intent(in)::data_loc
type(field_buffer_map_t),intent(inout)::field_buffers(:)
integer(8),intent(inout)::last_shape_dims(1_8:3_8)
integer(4),intent(inout)::last_stride_factors(1_8:3_8)
integer(8),intent(inout)::last_output_shape(1_8:3_8)
end
subroutine cleanup_field_buffers(field_buffers)
type(field_buffer_map_t),allocatable,intent(inout)::field_buffers(:)
end
end
