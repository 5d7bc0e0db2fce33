﻿!mod$ v1 sum:06d3c09989ba7313
!need$ fadd42cafe0c8e6b n m_io_session
!need$ f39a1ef65bd4689d n m_base_backend
!need$ f74ae58d325d162e n m_common
!need$ f4f3b1cdb42159bf n m_mesh
!need$ 939e7b51cda90705 n m_allocator
!need$ 0df96a70750958ab n mpi
!need$ f1de5abe9bfe2168 i iso_fortran_env
module m_ibm
use,intrinsic::iso_fortran_env,only:stderr=>error_unit
use m_io_session,only:reader_session_t
use m_allocator,only:allocator_t
use m_allocator,only:field_t
use m_base_backend,only:base_backend_t
use m_common,only:dp
use m_common,only:i8
use m_common,only:pi
use m_common,only:dir_x
use m_common,only:dir_c
use m_common,only:vert
use m_mesh,only:mesh_t
use mpi,only:mpi_source
use mpi,only:mpi_tag
use mpi,only:mpi_error
use mpi,only:mpi_status_size
use mpi,only:mpi_success
use mpi,only:mpi_err_other
use mpi,only:mpi_err_count
use mpi,only:mpi_err_spawn
use mpi,only:mpi_err_locktype
use mpi,only:mpi_err_op
use mpi,only:mpi_err_dup_datarep
use mpi,only:mpi_err_unsupported_datarep
use mpi,only:mpi_err_truncate
use mpi,only:mpi_err_info_nokey
use mpi,only:mpi_err_assert
use mpi,only:mpi_err_file_exists
use mpi,only:mpi_err_pending
use mpi,only:mpi_err_comm
use mpi,only:mpi_err_keyval
use mpi,only:mpi_err_name
use mpi,only:mpi_err_request
use mpi,only:mpi_err_type
use mpi,only:mpi_err_info_value
use mpi,only:mpi_err_rma_sync
use mpi,only:mpi_err_no_mem
use mpi,only:mpi_err_bad_file
use mpi,only:mpi_err_quota
use mpi,only:mpi_err_root
use mpi,only:mpi_err_service
use mpi,only:mpi_err_io
use mpi,only:mpi_err_rma_flavor
use mpi,only:mpi_err_access
use mpi,only:mpi_err_no_space
use mpi,only:mpi_err_conversion
use mpi,only:mpi_err_win
use mpi,only:mpi_err_file
use mpi,only:mpi_err_rma_shared
use mpi,only:mpi_err_base
use mpi,only:mpi_err_rma_conflict
use mpi,only:mpi_err_in_status
use mpi,only:mpi_err_info_key
use mpi,only:mpi_err_arg
use mpi,only:mpi_err_read_only
use mpi,only:mpi_err_size
use mpi,only:mpi_err_buffer
use mpi,only:mpi_err_lastcode
use mpi,only:mpi_err_disp
use mpi,only:mpi_err_port
use mpi,only:mpi_err_group
use mpi,only:mpi_err_topology
use mpi,only:mpi_err_tag
use mpi,only:mpi_err_not_same
use mpi,only:mpi_err_info
use mpi,only:mpi_err_unknown
use mpi,only:mpi_err_file_in_use
use mpi,only:mpi_err_rma_attach
use mpi,only:mpi_err_unsupported_operation
use mpi,only:mpi_err_amode
use mpi,only:mpi_err_rank
use mpi,only:mpi_err_dims
use mpi,only:mpi_err_no_such_file
use mpi,only:mpi_err_rma_range
use mpi,only:mpi_err_intern
use mpi,only:mpi_errors_are_fatal
use mpi,only:mpi_errors_return
use mpi,only:mpi_ident
use mpi,only:mpi_congruent
use mpi,only:mpi_similar
use mpi,only:mpi_unequal
use mpi,only:mpi_win_flavor_create
use mpi,only:mpi_win_flavor_allocate
use mpi,only:mpi_win_flavor_dynamic
use mpi,only:mpi_win_flavor_shared
use mpi,only:mpi_win_separate
use mpi,only:mpi_win_unified
use mpi,only:mpi_max
use mpi,only:mpi_min
use mpi,only:mpi_sum
use mpi,only:mpi_prod
use mpi,only:mpi_land
use mpi,only:mpi_band
use mpi,only:mpi_lor
use mpi,only:mpi_bor
use mpi,only:mpi_lxor
use mpi,only:mpi_bxor
use mpi,only:mpi_minloc
use mpi,only:mpi_maxloc
use mpi,only:mpi_replace
use mpi,only:mpi_no_op
use mpi,only:mpi_comm_world
use mpi,only:mpi_comm_self
use mpi,only:mpi_group_empty
use mpi,only:mpi_comm_null
use mpi,only:mpi_win_null
use mpi,only:mpi_file_null
use mpi,only:mpi_group_null
use mpi,only:mpi_op_null
use mpi,only:mpi_datatype_null
use mpi,only:mpi_request_null
use mpi,only:mpi_errhandler_null
use mpi,only:mpi_info_null
use mpi,only:mpi_info_env
use mpi,only:mpi_tag_ub
use mpi,only:mpi_host
use mpi,only:mpi_io
use mpi,only:mpi_wtime_is_global
use mpi,only:mpi_universe_size
use mpi,only:mpi_lastusedcode
use mpi,only:mpi_appnum
use mpi,only:mpi_win_base
use mpi,only:mpi_win_size
use mpi,only:mpi_win_disp_unit
use mpi,only:mpi_win_create_flavor
use mpi,only:mpi_win_model
use mpi,only:mpi_max_error_string
use mpi,only:mpi_max_port_name
use mpi,only:mpi_max_object_name
use mpi,only:mpi_max_info_key
use mpi,only:mpi_max_info_val
use mpi,only:mpi_max_processor_name
use mpi,only:mpi_max_datarep_string
use mpi,only:mpi_max_library_version_string
use mpi,only:mpi_undefined
use mpi,only:mpi_keyval_invalid
use mpi,only:mpi_bsend_overhead
use mpi,only:mpi_proc_null
use mpi,only:mpi_any_source
use mpi,only:mpi_any_tag
use mpi,only:mpi_root
use mpi,only:mpi_graph
use mpi,only:mpi_cart
use mpi,only:mpi_dist_graph
use mpi,only:mpi_version
use mpi,only:mpi_subversion
use mpi,only:mpi_lock_exclusive
use mpi,only:mpi_lock_shared
use mpi,only:mpi_complex
use mpi,only:mpi_double_complex
use mpi,only:mpi_logical
use mpi,only:mpi_real
use mpi,only:mpi_double_precision
use mpi,only:mpi_integer
use mpi,only:mpi_2integer
use mpi,only:mpi_2double_precision
use mpi,only:mpi_2real
use mpi,only:mpi_character
use mpi,only:mpi_byte
use mpi,only:mpi_ub
use mpi,only:mpi_lb
use mpi,only:mpi_packed
use mpi,only:mpi_integer1
use mpi,only:mpi_integer2
use mpi,only:mpi_integer4
use mpi,only:mpi_integer8
use mpi,only:mpi_integer16
use mpi,only:mpi_real4
use mpi,only:mpi_real8
use mpi,only:mpi_real16
use mpi,only:mpi_complex8
use mpi,only:mpi_complex16
use mpi,only:mpi_complex32
use mpi,only:mpi_address_kind
use mpi,only:mpi_offset_kind
use mpi,only:mpi_count_kind
use mpi,only:mpi_integer_kind
use mpi,only:mpi_char
use mpi,only:mpi_signed_char
use mpi,only:mpi_unsigned_char
use mpi,only:mpi_wchar
use mpi,only:mpi_short
use mpi,only:mpi_unsigned_short
use mpi,only:mpi_int
use mpi,only:mpi_unsigned
use mpi,only:mpi_long
use mpi,only:mpi_unsigned_long
use mpi,only:mpi_float
use mpi,only:mpi_double
use mpi,only:mpi_long_double
use mpi,only:mpi_long_long_int
use mpi,only:mpi_unsigned_long_long
use mpi,only:mpi_long_long
use mpi,only:mpi_float_int
use mpi,only:mpi_double_int
use mpi,only:mpi_long_int
use mpi,only:mpi_short_int
use mpi,only:mpi_2int
use mpi,only:mpi_long_double_int
use mpi,only:mpi_int8_t
use mpi,only:mpi_int16_t
use mpi,only:mpi_int32_t
use mpi,only:mpi_int64_t
use mpi,only:mpi_uint8_t
use mpi,only:mpi_uint16_t
use mpi,only:mpi_uint32_t
use mpi,only:mpi_uint64_t
use mpi,only:mpi_c_bool
use mpi,only:mpi_c_float_complex
use mpi,only:mpi_c_complex
use mpi,only:mpi_c_double_complex
use mpi,only:mpi_c_long_double_complex
use mpi,only:mpi_aint
use mpi,only:mpi_offset
use mpi,only:mpi_count
use mpi,only:mpi_cxx_bool
use mpi,only:mpi_cxx_float_complex
use mpi,only:mpi_cxx_double_complex
use mpi,only:mpi_cxx_long_double_complex
use mpi,only:mpi_combiner_named
use mpi,only:mpi_combiner_dup
use mpi,only:mpi_combiner_contiguous
use mpi,only:mpi_combiner_vector
use mpi,only:mpi_combiner_hvector_integer
use mpi,only:mpi_combiner_hvector
use mpi,only:mpi_combiner_indexed
use mpi,only:mpi_combiner_hindexed_integer
use mpi,only:mpi_combiner_hindexed
use mpi,only:mpi_combiner_indexed_block
use mpi,only:mpi_combiner_struct_integer
use mpi,only:mpi_combiner_struct
use mpi,only:mpi_combiner_subarray
use mpi,only:mpi_combiner_darray
use mpi,only:mpi_combiner_f90_real
use mpi,only:mpi_combiner_f90_complex
use mpi,only:mpi_combiner_f90_integer
use mpi,only:mpi_combiner_resized
use mpi,only:mpi_combiner_hindexed_block
use mpi,only:mpi_typeclass_real
use mpi,only:mpi_typeclass_integer
use mpi,only:mpi_typeclass_complex
use mpi,only:mpi_mode_nocheck
use mpi,only:mpi_mode_nostore
use mpi,only:mpi_mode_noput
use mpi,only:mpi_mode_noprecede
use mpi,only:mpi_mode_nosucceed
use mpi,only:mpi_comm_type_shared
use mpi,only:mpi_message_null
use mpi,only:mpi_message_no_proc
use mpi,only:mpi_thread_single
use mpi,only:mpi_thread_funneled
use mpi,only:mpi_thread_serialized
use mpi,only:mpi_thread_multiple
use mpi,only:mpi_mode_rdonly
use mpi,only:mpi_mode_rdwr
use mpi,only:mpi_mode_wronly
use mpi,only:mpi_mode_delete_on_close
use mpi,only:mpi_mode_unique_open
use mpi,only:mpi_mode_create
use mpi,only:mpi_mode_excl
use mpi,only:mpi_mode_append
use mpi,only:mpi_mode_sequential
use mpi,only:mpi_seek_set
use mpi,only:mpi_seek_cur
use mpi,only:mpi_seek_end
use mpi,only:mpi_order_c
use mpi,only:mpi_order_fortran
use mpi,only:mpi_distribute_block
use mpi,only:mpi_distribute_cyclic
use mpi,only:mpi_distribute_none
use mpi,only:mpi_distribute_dflt_darg
use mpi,only:mpi_displacement_current
use mpi,only:mpi_subarrays_supported
use mpi,only:mpi_async_protects_nonblocking
use mpi,only:mpi_dup_fn
use mpi,only:mpi_null_delete_fn
use mpi,only:mpi_null_copy_fn
use mpi,only:mpi_comm_dup_fn
use mpi,only:mpi_comm_null_delete_fn
use mpi,only:mpi_comm_null_copy_fn
use mpi,only:mpi_win_dup_fn
use mpi,only:mpi_win_null_delete_fn
use mpi,only:mpi_win_null_copy_fn
use mpi,only:mpi_type_dup_fn
use mpi,only:mpi_type_null_delete_fn
use mpi,only:mpi_type_null_copy_fn
use mpi,only:mpi_conversion_fn_null
use mpi,only:mpi_wtime
use mpi,only:mpi_wtick
use mpi,only:pmpi_wtime
use mpi,only:pmpi_wtick
use mpi,only:mpi_comm_rank
use mpi,only:mpi_comm_size
use mpi,only:mpi_abort
use mpi,only:mpi_reduce
use mpi,only:mpi_initialized
use mpi,only:mpi_unweighted
use mpi,only:mpi_weights_empty
use mpi,only:mpi_bottom
use mpi,only:mpi_in_place
use mpi,only:mpi_status_ignore
use mpi,only:mpi_statuses_ignore
use mpi,only:mpi_errcodes_ignore
use mpi,only:mpi_argvs_null
use mpi,only:mpi_argv_null
private::stderr
private::reader_session_t
private::allocator_t
private::field_t
private::base_backend_t
private::dp
private::i8
private::pi
private::dir_x
private::dir_c
private::vert
private::mesh_t
private::mpi_source
private::mpi_tag
private::mpi_error
private::mpi_status_size
private::mpi_success
private::mpi_err_other
private::mpi_err_count
private::mpi_err_spawn
private::mpi_err_locktype
private::mpi_err_op
private::mpi_err_dup_datarep
private::mpi_err_unsupported_datarep
private::mpi_err_truncate
private::mpi_err_info_nokey
private::mpi_err_assert
private::mpi_err_file_exists
private::mpi_err_pending
private::mpi_err_comm
private::mpi_err_keyval
private::mpi_err_name
private::mpi_err_request
private::mpi_err_type
private::mpi_err_info_value
private::mpi_err_rma_sync
private::mpi_err_no_mem
private::mpi_err_bad_file
private::mpi_err_quota
private::mpi_err_root
private::mpi_err_service
private::mpi_err_io
private::mpi_err_rma_flavor
private::mpi_err_access
private::mpi_err_no_space
private::mpi_err_conversion
private::mpi_err_win
private::mpi_err_file
private::mpi_err_rma_shared
private::mpi_err_base
private::mpi_err_rma_conflict
private::mpi_err_in_status
private::mpi_err_info_key
private::mpi_err_arg
private::mpi_err_read_only
private::mpi_err_size
private::mpi_err_buffer
private::mpi_err_lastcode
private::mpi_err_disp
private::mpi_err_port
private::mpi_err_group
private::mpi_err_topology
private::mpi_err_tag
private::mpi_err_not_same
private::mpi_err_info
private::mpi_err_unknown
private::mpi_err_file_in_use
private::mpi_err_rma_attach
private::mpi_err_unsupported_operation
private::mpi_err_amode
private::mpi_err_rank
private::mpi_err_dims
private::mpi_err_no_such_file
private::mpi_err_rma_range
private::mpi_err_intern
private::mpi_errors_are_fatal
private::mpi_errors_return
private::mpi_ident
private::mpi_congruent
private::mpi_similar
private::mpi_unequal
private::mpi_win_flavor_create
private::mpi_win_flavor_allocate
private::mpi_win_flavor_dynamic
private::mpi_win_flavor_shared
private::mpi_win_separate
private::mpi_win_unified
private::mpi_max
private::mpi_min
private::mpi_sum
private::mpi_prod
private::mpi_land
private::mpi_band
private::mpi_lor
private::mpi_bor
private::mpi_lxor
private::mpi_bxor
private::mpi_minloc
private::mpi_maxloc
private::mpi_replace
private::mpi_no_op
private::mpi_comm_world
private::mpi_comm_self
private::mpi_group_empty
private::mpi_comm_null
private::mpi_win_null
private::mpi_file_null
private::mpi_group_null
private::mpi_op_null
private::mpi_datatype_null
private::mpi_request_null
private::mpi_errhandler_null
private::mpi_info_null
private::mpi_info_env
private::mpi_tag_ub
private::mpi_host
private::mpi_io
private::mpi_wtime_is_global
private::mpi_universe_size
private::mpi_lastusedcode
private::mpi_appnum
private::mpi_win_base
private::mpi_win_size
private::mpi_win_disp_unit
private::mpi_win_create_flavor
private::mpi_win_model
private::mpi_max_error_string
private::mpi_max_port_name
private::mpi_max_object_name
private::mpi_max_info_key
private::mpi_max_info_val
private::mpi_max_processor_name
private::mpi_max_datarep_string
private::mpi_max_library_version_string
private::mpi_undefined
private::mpi_keyval_invalid
private::mpi_bsend_overhead
private::mpi_proc_null
private::mpi_any_source
private::mpi_any_tag
private::mpi_root
private::mpi_graph
private::mpi_cart
private::mpi_dist_graph
private::mpi_version
private::mpi_subversion
private::mpi_lock_exclusive
private::mpi_lock_shared
private::mpi_complex
private::mpi_double_complex
private::mpi_logical
private::mpi_real
private::mpi_double_precision
private::mpi_integer
private::mpi_2integer
private::mpi_2double_precision
private::mpi_2real
private::mpi_character
private::mpi_byte
private::mpi_ub
private::mpi_lb
private::mpi_packed
private::mpi_integer1
private::mpi_integer2
private::mpi_integer4
private::mpi_integer8
private::mpi_integer16
private::mpi_real4
private::mpi_real8
private::mpi_real16
private::mpi_complex8
private::mpi_complex16
private::mpi_complex32
private::mpi_address_kind
private::mpi_offset_kind
private::mpi_count_kind
private::mpi_integer_kind
private::mpi_char
private::mpi_signed_char
private::mpi_unsigned_char
private::mpi_wchar
private::mpi_short
private::mpi_unsigned_short
private::mpi_int
private::mpi_unsigned
private::mpi_long
private::mpi_unsigned_long
private::mpi_float
private::mpi_double
private::mpi_long_double
private::mpi_long_long_int
private::mpi_unsigned_long_long
private::mpi_long_long
private::mpi_float_int
private::mpi_double_int
private::mpi_long_int
private::mpi_short_int
private::mpi_2int
private::mpi_long_double_int
private::mpi_int8_t
private::mpi_int16_t
private::mpi_int32_t
private::mpi_int64_t
private::mpi_uint8_t
private::mpi_uint16_t
private::mpi_uint32_t
private::mpi_uint64_t
private::mpi_c_bool
private::mpi_c_float_complex
private::mpi_c_complex
private::mpi_c_double_complex
private::mpi_c_long_double_complex
private::mpi_aint
private::mpi_offset
private::mpi_count
private::mpi_cxx_bool
private::mpi_cxx_float_complex
private::mpi_cxx_double_complex
private::mpi_cxx_long_double_complex
private::mpi_combiner_named
private::mpi_combiner_dup
private::mpi_combiner_contiguous
private::mpi_combiner_vector
private::mpi_combiner_hvector_integer
private::mpi_combiner_hvector
private::mpi_combiner_indexed
private::mpi_combiner_hindexed_integer
private::mpi_combiner_hindexed
private::mpi_combiner_indexed_block
private::mpi_combiner_struct_integer
private::mpi_combiner_struct
private::mpi_combiner_subarray
private::mpi_combiner_darray
private::mpi_combiner_f90_real
private::mpi_combiner_f90_complex
private::mpi_combiner_f90_integer
private::mpi_combiner_resized
private::mpi_combiner_hindexed_block
private::mpi_typeclass_real
private::mpi_typeclass_integer
private::mpi_typeclass_complex
private::mpi_mode_nocheck
private::mpi_mode_nostore
private::mpi_mode_noput
private::mpi_mode_noprecede
private::mpi_mode_nosucceed
private::mpi_comm_type_shared
private::mpi_message_null
private::mpi_message_no_proc
private::mpi_thread_single
private::mpi_thread_funneled
private::mpi_thread_serialized
private::mpi_thread_multiple
private::mpi_mode_rdonly
private::mpi_mode_rdwr
private::mpi_mode_wronly
private::mpi_mode_delete_on_close
private::mpi_mode_unique_open
private::mpi_mode_create
private::mpi_mode_excl
private::mpi_mode_append
private::mpi_mode_sequential
private::mpi_seek_set
private::mpi_seek_cur
private::mpi_seek_end
private::mpi_order_c
private::mpi_order_fortran
private::mpi_distribute_block
private::mpi_distribute_cyclic
private::mpi_distribute_none
private::mpi_distribute_dflt_darg
private::mpi_displacement_current
private::mpi_subarrays_supported
private::mpi_async_protects_nonblocking
private::mpi_dup_fn
private::mpi_null_delete_fn
private::mpi_null_copy_fn
private::mpi_comm_dup_fn
private::mpi_comm_null_delete_fn
private::mpi_comm_null_copy_fn
private::mpi_win_dup_fn
private::mpi_win_null_delete_fn
private::mpi_win_null_copy_fn
private::mpi_type_dup_fn
private::mpi_type_null_delete_fn
private::mpi_type_null_copy_fn
private::mpi_conversion_fn_null
private::mpi_wtime
private::mpi_wtick
private::pmpi_wtime
private::pmpi_wtick
private::mpi_comm_rank
private::mpi_comm_size
private::mpi_abort
private::mpi_reduce
private::mpi_initialized
private::mpi_unweighted
private::mpi_weights_empty
private::mpi_bottom
private::mpi_in_place
private::mpi_status_ignore
private::mpi_statuses_ignore
private::mpi_errcodes_ignore
private::mpi_argvs_null
private::mpi_argv_null
integer(4),parameter::iibm_basic=1_4
type::ibm_t
class(base_backend_t),pointer::backend=>NULL()
class(mesh_t),pointer::mesh=>NULL()
type(allocator_t),pointer::host_allocator=>NULL()
integer(4)::iibm=0_4
class(field_t),pointer::ep1=>NULL()
contains
procedure::body
end type
intrinsic::null
private::null
private::init
private::body
interface ibm_t
procedure::init
end interface
contains
function init(backend,mesh,host_allocator) result(ibm)
class(base_backend_t),intent(inout),target::backend
type(mesh_t),intent(inout),target::mesh
type(allocator_t),intent(inout),target::host_allocator
type(ibm_t)::ibm
end
subroutine body(self,u,v,w)
class(ibm_t)::self
class(field_t),intent(inout)::u
class(field_t),intent(inout)::v
class(field_t),intent(inout)::w
end
end
