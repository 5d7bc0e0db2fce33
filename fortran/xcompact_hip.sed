# The HIP branch of the reference's main program, as an EDIT of /root/reference/src/xcompact.f90 that `make -C fortran`
# applies where the reference tree is mounted (the result is compiled from a temporary file and never kept): the
# reference selects its backend at compile time with an #ifdef CUDA branch (src/xcompact.f90:15-22, 32-38, 57-64,
# 87-107); a third backend is that branch with the names of fortran/m_hip_backend.f90.  This file IS the patch a
# maintainer would apply (INTEGRATION.md) -- everything it does not touch is the reference's own program.
#
# the compile-time switch
s/^#ifdef CUDA/#ifdef HIP/
# modules (src/xcompact.f90:16-18)
s/^  use m_cuda_allocator$/  use m_hip_allocator, only: hip_allocator_t, hip_allocator_init/
s/^  use m_cuda_backend$/  use m_hip_backend, only: hip_backend_t, hip_backend_init/
s/^  use m_cuda_common, only: SZ$/  use m_hip_common, only: SZ\n  use m_x3d2_hip_capi, only: x3d_device_count, x3d_check/
# one rank <-> one device, round robin (src/xcompact.f90:57-60): the device count comes from the library, the device
# is handed to the allocator's constructor, which creates the library context on it
s/^  ierr = cudaGetDeviceCount(ndevs)$/  call x3d_check(x3d_device_count(ndevs))/
s/^  ierr = cudaSetDevice(mod(nrank, ndevs)) ! round-robin$/  devnum = mod(nrank, max(ndevs, 1)) ! round-robin/
/^  ierr = cudaGetDevice(devnum)$/d
s/backend_name = "CUDA"/backend_name = "HIP"/
# constructors (src/xcompact.f90:88, 95)
s/cuda_allocator_t(dims, SZ)/hip_allocator_init(dims, SZ, devnum)/
s/cuda_backend_t(mesh, allocator)/hip_backend_init(mesh, allocator)/
# types, variables, messages
s/cuda_backend_t/hip_backend_t/g
s/cuda_allocator_t/hip_allocator_t/g
s/cuda_backend/hip_backend/g
s/cuda_allocator/hip_allocator/g
s/'CUDA allocator instantiated'/'HIP allocator instantiated, rank', nrank, 'on device', devnum/
s/if (nrank == 0) print \*, 'HIP allocator instantiated/print *, 'HIP allocator instantiated/
s/'CUDA backend instantiated'/'HIP (MI355X) backend instantiated'/
