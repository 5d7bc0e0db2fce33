! Build glue for the reference-oracle build (test infrastructure, never shipped).
! The image's MPICH (/opt/conda) ships `mpif.h` and `libmpifort.so`, but its
! `mpi.mod` is in gfortran's module format, which flang cannot read.  This file
! re-exports the image's own `mpif.h` as `module mpi` so the reference's
! `use mpi` resolves against the REAL MPI library present in the image.
module mpi
  implicit none
  include 'mpif.h'
  ! names the reference imports with `use mpi, only: ...`; the bodies are the
  ! real MPICH routines in libmpifort.so (mpif.h itself declares only
  ! MPI_WTIME/MPI_WTICK/PMPI_* as external).
  external :: MPI_Comm_rank, MPI_Comm_size, MPI_Abort, MPI_Reduce, MPI_Initialized
end module mpi
