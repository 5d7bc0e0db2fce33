﻿!mod$ v1 sum:0bc2350bced86eca
!need$ 99145601f71fb607 n m_base_case
!need$ f74ae58d325d162e n m_common
!need$ 7f5e804034ee5163 n m_config
!need$ d9a8bda24462498c n m_field
!need$ f4f3b1cdb42159bf n m_mesh
!need$ 85f841a7a38b0974 n m_solver
!need$ 0df96a70750958ab n mpi
!need$ f1de5abe9bfe2168 i iso_fortran_env
!need$ 939e7b51cda90705 n m_allocator
!need$ f39a1ef65bd4689d n m_base_backend
module m_case_channel
use,intrinsic::iso_fortran_env,only:stderr=>error_unit
use m_allocator,only:allocator_t
use m_base_backend,only:base_backend_t
use m_base_case,only:base_case_t
use m_common,only:dp
use m_common,only:mpi_x3d2_dp
use m_common,only:get_argument
use m_common,only:dir_c
use m_common,only:dir_x
use m_common,only:vert
use m_common,only:cell
use m_common,only:y_face
use m_common,only:bc_dirichlet
use m_config,only:channel_config_t
use m_field,only:field_t
use m_mesh,only:mesh_t
use m_solver,only:init
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
type,extends(base_case_t)::case_channel_t
type(channel_config_t)::channel_cfg
contains
procedure::define_bc=>define_bc_channel
procedure::initial_conditions=>initial_conditions_channel
procedure::forcings=>forcings_channel
procedure::apply_bc=>apply_bc_channel
procedure::postprocess=>postprocess_channel
end type
interface case_channel_t
procedure::case_channel_init
end interface
contains
function case_channel_init(backend,mesh,host_allocator) result(flow_case)
class(base_backend_t),intent(inout),target::backend
type(mesh_t),intent(inout),target::mesh
type(allocator_t),intent(inout),target::host_allocator
type(case_channel_t)::flow_case
end
subroutine define_bc_channel(self)
class(case_channel_t)::self
end
subroutine initial_conditions_channel(self)
class(case_channel_t)::self
end
subroutine forcings_channel(self,du,dv,dw,iter)
class(case_channel_t)::self
class(field_t),intent(inout)::du
class(field_t),intent(inout)::dv
class(field_t),intent(inout)::dw
integer(4),intent(in)::iter
end
subroutine apply_bc_channel(self,u,v,w)
class(case_channel_t)::self
class(field_t),intent(inout)::u
class(field_t),intent(inout)::v
class(field_t),intent(inout)::w
end
subroutine postprocess_channel(self,iter,t)
class(case_channel_t)::self
integer(4),intent(in)::iter
real(8),intent(in)::t
end
end
