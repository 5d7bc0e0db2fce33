﻿!mod$ v1 sum:7f5e804034ee5163
!need$ f1de5abe9bfe2168 i iso_fortran_env
!need$ f74ae58d325d162e n m_common
module m_config
use,intrinsic::iso_fortran_env,only:stderr=>error_unit
use m_common,only:mpi_source
use m_common,only:mpi_tag
use m_common,only:mpi_error
use m_common,only:mpi_status_size
use m_common,only:mpi_success
use m_common,only:mpi_err_other
use m_common,only:mpi_err_count
use m_common,only:mpi_err_spawn
use m_common,only:mpi_err_locktype
use m_common,only:mpi_err_op
use m_common,only:mpi_err_dup_datarep
use m_common,only:mpi_err_unsupported_datarep
use m_common,only:mpi_err_truncate
use m_common,only:mpi_err_info_nokey
use m_common,only:mpi_err_assert
use m_common,only:mpi_err_file_exists
use m_common,only:mpi_err_pending
use m_common,only:mpi_err_comm
use m_common,only:mpi_err_keyval
use m_common,only:mpi_err_name
use m_common,only:mpi_err_request
use m_common,only:mpi_err_type
use m_common,only:mpi_err_info_value
use m_common,only:mpi_err_rma_sync
use m_common,only:mpi_err_no_mem
use m_common,only:mpi_err_bad_file
use m_common,only:mpi_err_quota
use m_common,only:mpi_err_root
use m_common,only:mpi_err_service
use m_common,only:mpi_err_io
use m_common,only:mpi_err_rma_flavor
use m_common,only:mpi_err_access
use m_common,only:mpi_err_no_space
use m_common,only:mpi_err_conversion
use m_common,only:mpi_err_win
use m_common,only:mpi_err_file
use m_common,only:mpi_err_rma_shared
use m_common,only:mpi_err_base
use m_common,only:mpi_err_rma_conflict
use m_common,only:mpi_err_in_status
use m_common,only:mpi_err_info_key
use m_common,only:mpi_err_arg
use m_common,only:mpi_err_read_only
use m_common,only:mpi_err_size
use m_common,only:mpi_err_buffer
use m_common,only:mpi_err_lastcode
use m_common,only:mpi_err_disp
use m_common,only:mpi_err_port
use m_common,only:mpi_err_group
use m_common,only:mpi_err_topology
use m_common,only:mpi_err_tag
use m_common,only:mpi_err_not_same
use m_common,only:mpi_err_info
use m_common,only:mpi_err_unknown
use m_common,only:mpi_err_file_in_use
use m_common,only:mpi_err_rma_attach
use m_common,only:mpi_err_unsupported_operation
use m_common,only:mpi_err_amode
use m_common,only:mpi_err_rank
use m_common,only:mpi_err_dims
use m_common,only:mpi_err_no_such_file
use m_common,only:mpi_err_rma_range
use m_common,only:mpi_err_intern
use m_common,only:mpi_errors_are_fatal
use m_common,only:mpi_errors_return
use m_common,only:mpi_ident
use m_common,only:mpi_congruent
use m_common,only:mpi_similar
use m_common,only:mpi_unequal
use m_common,only:mpi_win_flavor_create
use m_common,only:mpi_win_flavor_allocate
use m_common,only:mpi_win_flavor_dynamic
use m_common,only:mpi_win_flavor_shared
use m_common,only:mpi_win_separate
use m_common,only:mpi_win_unified
use m_common,only:mpi_max
use m_common,only:mpi_min
use m_common,only:mpi_sum
use m_common,only:mpi_prod
use m_common,only:mpi_land
use m_common,only:mpi_band
use m_common,only:mpi_lor
use m_common,only:mpi_bor
use m_common,only:mpi_lxor
use m_common,only:mpi_bxor
use m_common,only:mpi_minloc
use m_common,only:mpi_maxloc
use m_common,only:mpi_replace
use m_common,only:mpi_no_op
use m_common,only:mpi_comm_world
use m_common,only:mpi_comm_self
use m_common,only:mpi_group_empty
use m_common,only:mpi_comm_null
use m_common,only:mpi_win_null
use m_common,only:mpi_file_null
use m_common,only:mpi_group_null
use m_common,only:mpi_op_null
use m_common,only:mpi_datatype_null
use m_common,only:mpi_request_null
use m_common,only:mpi_errhandler_null
use m_common,only:mpi_info_null
use m_common,only:mpi_info_env
use m_common,only:mpi_tag_ub
use m_common,only:mpi_host
use m_common,only:mpi_io
use m_common,only:mpi_wtime_is_global
use m_common,only:mpi_universe_size
use m_common,only:mpi_lastusedcode
use m_common,only:mpi_appnum
use m_common,only:mpi_win_base
use m_common,only:mpi_win_size
use m_common,only:mpi_win_disp_unit
use m_common,only:mpi_win_create_flavor
use m_common,only:mpi_win_model
use m_common,only:mpi_max_error_string
use m_common,only:mpi_max_port_name
use m_common,only:mpi_max_object_name
use m_common,only:mpi_max_info_key
use m_common,only:mpi_max_info_val
use m_common,only:mpi_max_processor_name
use m_common,only:mpi_max_datarep_string
use m_common,only:mpi_max_library_version_string
use m_common,only:mpi_undefined
use m_common,only:mpi_keyval_invalid
use m_common,only:mpi_bsend_overhead
use m_common,only:mpi_proc_null
use m_common,only:mpi_any_source
use m_common,only:mpi_any_tag
use m_common,only:mpi_root
use m_common,only:mpi_graph
use m_common,only:mpi_cart
use m_common,only:mpi_dist_graph
use m_common,only:mpi_version
use m_common,only:mpi_subversion
use m_common,only:mpi_lock_exclusive
use m_common,only:mpi_lock_shared
use m_common,only:mpi_complex
use m_common,only:mpi_double_complex
use m_common,only:mpi_logical
use m_common,only:mpi_real
use m_common,only:mpi_double_precision
use m_common,only:mpi_integer
use m_common,only:mpi_2integer
use m_common,only:mpi_2double_precision
use m_common,only:mpi_2real
use m_common,only:mpi_character
use m_common,only:mpi_byte
use m_common,only:mpi_ub
use m_common,only:mpi_lb
use m_common,only:mpi_packed
use m_common,only:mpi_integer1
use m_common,only:mpi_integer2
use m_common,only:mpi_integer4
use m_common,only:mpi_integer8
use m_common,only:mpi_integer16
use m_common,only:mpi_real4
use m_common,only:mpi_real8
use m_common,only:mpi_real16
use m_common,only:mpi_complex8
use m_common,only:mpi_complex16
use m_common,only:mpi_complex32
use m_common,only:mpi_address_kind
use m_common,only:mpi_offset_kind
use m_common,only:mpi_count_kind
use m_common,only:mpi_integer_kind
use m_common,only:mpi_char
use m_common,only:mpi_signed_char
use m_common,only:mpi_unsigned_char
use m_common,only:mpi_wchar
use m_common,only:mpi_short
use m_common,only:mpi_unsigned_short
use m_common,only:mpi_int
use m_common,only:mpi_unsigned
use m_common,only:mpi_long
use m_common,only:mpi_unsigned_long
use m_common,only:mpi_float
use m_common,only:mpi_double
use m_common,only:mpi_long_double
use m_common,only:mpi_long_long_int
use m_common,only:mpi_unsigned_long_long
use m_common,only:mpi_long_long
use m_common,only:mpi_float_int
use m_common,only:mpi_double_int
use m_common,only:mpi_long_int
use m_common,only:mpi_short_int
use m_common,only:mpi_2int
use m_common,only:mpi_long_double_int
use m_common,only:mpi_int8_t
use m_common,only:mpi_int16_t
use m_common,only:mpi_int32_t
use m_common,only:mpi_int64_t
use m_common,only:mpi_uint8_t
use m_common,only:mpi_uint16_t
use m_common,only:mpi_uint32_t
use m_common,only:mpi_uint64_t
use m_common,only:mpi_c_bool
use m_common,only:mpi_c_float_complex
use m_common,only:mpi_c_complex
use m_common,only:mpi_c_double_complex
use m_common,only:mpi_c_long_double_complex
use m_common,only:mpi_aint
use m_common,only:mpi_offset
use m_common,only:mpi_count
use m_common,only:mpi_cxx_bool
use m_common,only:mpi_cxx_float_complex
use m_common,only:mpi_cxx_double_complex
use m_common,only:mpi_cxx_long_double_complex
use m_common,only:mpi_combiner_named
use m_common,only:mpi_combiner_dup
use m_common,only:mpi_combiner_contiguous
use m_common,only:mpi_combiner_vector
use m_common,only:mpi_combiner_hvector_integer
use m_common,only:mpi_combiner_hvector
use m_common,only:mpi_combiner_indexed
use m_common,only:mpi_combiner_hindexed_integer
use m_common,only:mpi_combiner_hindexed
use m_common,only:mpi_combiner_indexed_block
use m_common,only:mpi_combiner_struct_integer
use m_common,only:mpi_combiner_struct
use m_common,only:mpi_combiner_subarray
use m_common,only:mpi_combiner_darray
use m_common,only:mpi_combiner_f90_real
use m_common,only:mpi_combiner_f90_complex
use m_common,only:mpi_combiner_f90_integer
use m_common,only:mpi_combiner_resized
use m_common,only:mpi_combiner_hindexed_block
use m_common,only:mpi_typeclass_real
use m_common,only:mpi_typeclass_integer
use m_common,only:mpi_typeclass_complex
use m_common,only:mpi_mode_nocheck
use m_common,only:mpi_mode_nostore
use m_common,only:mpi_mode_noput
use m_common,only:mpi_mode_noprecede
use m_common,only:mpi_mode_nosucceed
use m_common,only:mpi_comm_type_shared
use m_common,only:mpi_message_null
use m_common,only:mpi_message_no_proc
use m_common,only:mpi_thread_single
use m_common,only:mpi_thread_funneled
use m_common,only:mpi_thread_serialized
use m_common,only:mpi_thread_multiple
use m_common,only:mpi_mode_rdonly
use m_common,only:mpi_mode_rdwr
use m_common,only:mpi_mode_wronly
use m_common,only:mpi_mode_delete_on_close
use m_common,only:mpi_mode_unique_open
use m_common,only:mpi_mode_create
use m_common,only:mpi_mode_excl
use m_common,only:mpi_mode_append
use m_common,only:mpi_mode_sequential
use m_common,only:mpi_seek_set
use m_common,only:mpi_seek_cur
use m_common,only:mpi_seek_end
use m_common,only:mpi_order_c
use m_common,only:mpi_order_fortran
use m_common,only:mpi_distribute_block
use m_common,only:mpi_distribute_cyclic
use m_common,only:mpi_distribute_none
use m_common,only:mpi_distribute_dflt_darg
use m_common,only:mpi_displacement_current
use m_common,only:mpi_subarrays_supported
use m_common,only:mpi_async_protects_nonblocking
use m_common,only:mpi_dup_fn
use m_common,only:mpi_null_delete_fn
use m_common,only:mpi_null_copy_fn
use m_common,only:mpi_comm_dup_fn
use m_common,only:mpi_comm_null_delete_fn
use m_common,only:mpi_comm_null_copy_fn
use m_common,only:mpi_win_dup_fn
use m_common,only:mpi_win_null_delete_fn
use m_common,only:mpi_win_null_copy_fn
use m_common,only:mpi_type_dup_fn
use m_common,only:mpi_type_null_delete_fn
use m_common,only:mpi_type_null_copy_fn
use m_common,only:mpi_conversion_fn_null
use m_common,only:mpi_wtime
use m_common,only:mpi_wtick
use m_common,only:pmpi_wtime
use m_common,only:pmpi_wtick
use m_common,only:mpi_comm_rank
use m_common,only:mpi_comm_size
use m_common,only:mpi_abort
use m_common,only:mpi_reduce
use m_common,only:mpi_initialized
use m_common,only:mpi_unweighted
use m_common,only:mpi_weights_empty
use m_common,only:mpi_bottom
use m_common,only:mpi_in_place
use m_common,only:mpi_status_ignore
use m_common,only:mpi_statuses_ignore
use m_common,only:mpi_errcodes_ignore
use m_common,only:mpi_argvs_null
use m_common,only:mpi_argv_null
use m_common,only:dp
use m_common,only:kind
use m_common,only:nbytes
use m_common,only:mpi_x3d2_dp
use m_common,only:is_sp
use m_common,only:sp
use m_common,only:i8
use m_common,only:selected_int_kind
use m_common,only:pi
use m_common,only:atan
use m_common,only:rdr_x2y
use m_common,only:rdr_x2z
use m_common,only:rdr_y2x
use m_common,only:rdr_y2z
use m_common,only:rdr_z2x
use m_common,only:rdr_z2y
use m_common,only:rdr_c2x
use m_common,only:rdr_c2y
use m_common,only:rdr_c2z
use m_common,only:rdr_x2c
use m_common,only:rdr_y2c
use m_common,only:rdr_z2c
use m_common,only:dir_x
use m_common,only:dir_y
use m_common,only:dir_z
use m_common,only:dir_c
use m_common,only:poisson_solver_fft
use m_common,only:poisson_solver_cg
use m_common,only:vert
use m_common,only:cell
use m_common,only:x_face
use m_common,only:y_face
use m_common,only:z_face
use m_common,only:x_edge
use m_common,only:y_edge
use m_common,only:z_edge
use m_common,only:null_loc
use m_common,only:bc_periodic
use m_common,only:bc_neumann
use m_common,only:bc_dirichlet
use m_common,only:bc_halo
use m_common,only:rdr_map
use m_common,only:reshape
use m_common,only:get_dirs_from_rdr
use m_common,only:get_rdr_from_dirs
use m_common,only:get_argument
use m_common,only:move_data_loc
integer(4),parameter::n_species_max=99_4
integer(4),parameter::max_output_fields=10_4
type,abstract::base_config_t
contains
procedure(read),deferred::read
end type
type,extends(base_config_t)::domain_config_t
character(30_4,1)::flow_case_name
real(8)::l_global(1_8:3_8)
integer(4)::dims_global(1_8:3_8)
integer(4)::nproc_dir(1_8:3_8)
character(20_4,1)::bc_x(1_8:2_8)
character(20_4,1)::bc_y(1_8:2_8)
character(20_4,1)::bc_z(1_8:2_8)
character(20_4,1)::stretching(1_8:3_8)
real(8)::beta(1_8:3_8)
contains
procedure::read=>read_domain_nml
end type
type,extends(base_config_t)::solver_config_t
real(8)::re
real(8)::dt
logical(4)::ibm_on
real(8),allocatable::pr_species(:)
integer(4)::n_iters
integer(4)::n_output
integer(4)::n_species
logical(4)::lowmem_transeq
logical(4)::lowmem_fft
character(3_4,1)::poisson_solver_type
character(3_4,1)::time_intg
character(30_4,1)::der1st_scheme
character(30_4,1)::der2nd_scheme
character(30_4,1)::interpl_scheme
character(30_4,1)::stagder_scheme
contains
procedure::read=>read_solver_nml
end type
type,extends(base_config_t)::channel_config_t
real(8)::omega_rot
real(8)::init_noise(1_8:3_8)
real(8)::inlet_noise(1_8:3_8)
logical(4)::rotation
integer(4)::n_rotate
contains
procedure::read=>read_channel_nml
end type
type,extends(base_config_t)::cylinder_config_t
real(8)::init_noise(1_8:3_8)
real(8)::inlet_noise(1_8:3_8)
contains
procedure::read=>read_cylinder_nml
end type
type,extends(base_config_t)::stats_config_t
integer(4)::initstat=0_4
integer(4)::istatfreq=1_4
integer(4)::istatout=0_4
character(256_4,1)::stats_prefix="statistics                                                                                                                                                                                                                                                      "
contains
procedure::read=>read_stats_nml
end type
type,extends(base_config_t)::checkpoint_config_t
integer(4)::checkpoint_freq=0_4
integer(4)::snapshot_freq=0_4
logical(4)::keep_checkpoint=.true._4
character(256_4,1)::checkpoint_prefix="checkpoint                                                                                                                                                                                                                                                      "
character(256_4,1)::snapshot_prefix="snapshot                                                                                                                                                                                                                                                        "
logical(4)::restart_from_checkpoint=.false._4
character(256_4,1)::restart_file="                                                                                                                                                                                                                                                                "
integer(4)::output_stride(1_8:3_8)=[INTEGER(4)::2_4,2_4,2_4]
logical(4)::snapshot_sp=.false._4
character(32_4,1)::output_fields(1_8:10_8)=[CHARACTER(KIND=1,LEN=32)::"                                ","                                ","                                ","                                ","                                ","                                ","                                ","                                ","                                ","                                "]
contains
procedure::read=>read_checkpoint_nml
end type
abstract interface
subroutine read(self,nml_file,nml_string)
import::base_config_t
class(base_config_t)::self
character(*,1),intent(in),optional::nml_file
character(*,1),intent(in),optional::nml_string
end
end interface
contains
subroutine read_domain_nml(self,nml_file,nml_string)
class(domain_config_t)::self
character(*,1),intent(in),optional::nml_file
character(*,1),intent(in),optional::nml_string
end
subroutine read_solver_nml(self,nml_file,nml_string)
class(solver_config_t)::self
character(*,1),intent(in),optional::nml_file
character(*,1),intent(in),optional::nml_string
end
subroutine read_channel_nml(self,nml_file,nml_string)
class(channel_config_t)::self
character(*,1),intent(in),optional::nml_file
character(*,1),intent(in),optional::nml_string
end
subroutine read_cylinder_nml(self,nml_file,nml_string)
class(cylinder_config_t)::self
character(*,1),intent(in),optional::nml_file
character(*,1),intent(in),optional::nml_string
end
subroutine read_checkpoint_nml(self,nml_file,nml_string)
class(checkpoint_config_t)::self
character(*,1),intent(in),optional::nml_file
character(*,1),intent(in),optional::nml_string
end
subroutine read_stats_nml(self,nml_file,nml_string)
class(stats_config_t)::self
character(*,1),intent(in),optional::nml_file
character(*,1),intent(in),optional::nml_string
end
pure function has_output_field(config,name)
type(checkpoint_config_t),intent(in)::config
character(*,1),intent(in)::name
logical(4)::has_output_field
end
end
