/* foamyade_mpi.h -- optional MPI implementation of fy_transport (libfoamyade_mpi.so).
 *
 * libfoamyade_hip.so itself never links an MPI: the wire protocol of FoamYade.C goes through the fy_transport
 * callbacks.  This small companion library implements them with the MPI the host application already uses, and performs the
 * communicator split the reference gets from its patched OpenFOAM Pstream (PstreamGlobals::MPI_COMM_FOAM, FoamYade.C:4,21-22):
 * Yade ranks come first in MPI_COMM_WORLD (README.md:29), solver ranks after them.
 */
#ifndef FOAMYADE_MPI_H
#define FOAMYADE_MPI_H
#include "foamyade_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* MPI must already be initialised.  n_yade_ranks = number of leading world ranks that belong to Yade (commSzDff, FoamYade.C:28).
 * Collective over MPI_COMM_WORLD (it calls MPI_Comm_split), so Yade-side ranks must make the matching split themselves, as
 * they do for the reference.  Fills *out; returns FY_OK or FY_ERR_TRANSPORT. */
int fy_mpi_transport_create(int n_yade_ranks, fy_transport* out);
int fy_mpi_transport_destroy(fy_transport* t);
/* the solver ranks' communicator the split produced (what OpenFOAM's -parallel run has as its world): *mpi_comm_out is an MPI_Comm */
int fy_mpi_local_comm(const fy_transport* t, void* mpi_comm_out);
/* a z-slab communicator (fy_solver_create_slab) over the ranks of *mpi_comm (an MPI_Comm; collective over it):
 *   use_rccl != 0  one GPU per rank -- halos, reductions and the coarse-level gather run over RCCL / xGMI, MPI only distributes the communicator id;
 *   use_rccl == 0  ranks that share a GPU (or no RCCL): the library stages the planes through pinned host memory and MPI moves them */
int fy_mpi_comm_create(const void* mpi_comm, int use_rccl, int device_ordinal, fy_comm** out);
#ifdef __cplusplus
}
#endif
#endif
