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
/* MPI must already be initialised.  n_yade_ranks = number of leading world ranks that belong to Yade (commSzDff, FoamYade.C:28); < 0: derive it
 * as the reference does -- every caller is a solver rank, the Yade ranks are the world's other ranks (world size - solver communicator size).
 * Collective over MPI_COMM_WORLD (it calls MPI_Comm_split), so Yade-side ranks must make the matching split themselves, as
 * they do for the reference.  Fills *out (zeroing it first: the optional callbacks stay null); returns FY_OK or FY_ERR_TRANSPORT.
 * n_yade_ranks < 0 in a world WITHOUT Yade ranks: FY_OK with out->send == NULL -- nobody to couple with; pass a NULL transport to the library
 * (fy_mpi_local_comm still returns the solver ranks' communicator).
 * Lifetime: destroy the objects that hold the transport (fy_destroy / fy_solver_destroy / fy_ldu_solver_destroy) BEFORE fy_mpi_transport_destroy:
 * they may have page-locked the wire helpers' arena and release it on destruction. */
int fy_mpi_transport_create(int n_yade_ranks, fy_transport* out);
int fy_mpi_transport_destroy(fy_transport* t);
/* WIRE HELPERS (round 4): K solver-side MPI ranks in front of ONE GPU.  One receiving core copies ~9 GB/s out of the MPI library whatever it
 * posts (tools/native/mpi_recv_rate.cpp: blocking receives, MPI_Irecv x 4 and four threads under MPI_THREAD_MULTIPLE all give the same rate),
 * so the 1.3 GB a 10 M-particle step moves across the wire need several receiving PROCESSES -- which is also how the reference scales it:
 * N solver ranks, each answering for the particles inside the bounding box it announced (FoamYade.C:77-155).  Here the first solver-side rank
 * computes (it announces a box no particle can touch) and the K - 1 others are helpers: each announces a z-slab of the block, receives that
 * slab's records from every Yade worker straight into a shared-memory arena, and sends the found flags and forces back out of it; the
 * computing rank hands the library VIEWS of the arena (fy_transport::recv_view / send_reserve / send_commit), so the PCIe copies start from
 * and end in it and the computing rank itself moves no payload.  To Yade this is a K-rank solver; the results are those of one rank (a record
 * that reaches two helpers is located by exactly one: fy_wire_pieces).
 * Collective over MPI_COMM_WORLD like fy_mpi_transport_create.  *is_helper = 0: *out is the computing rank's transport (use it as usual);
 * *is_helper = 1: call fy_mpi_wire_helper_serve(out), which returns when the computing rank destroys its transport.  10-double records only. */
int fy_mpi_transport_create_wire_helpers(int n_yade_ranks, fy_transport* out, int* is_helper);
int fy_mpi_wire_helper_serve(fy_transport* t);
/* the solver ranks' communicator the split produced (what OpenFOAM's -parallel run has as its world): *mpi_comm_out is an MPI_Comm */
int fy_mpi_local_comm(const fy_transport* t, void* mpi_comm_out);
/* a z-slab communicator (fy_solver_create_slab) over the ranks of *mpi_comm (an MPI_Comm; collective over it):
 *   use_rccl != 0  one GPU per rank -- halos, reductions and the coarse-level gather run over RCCL / xGMI, MPI only distributes the communicator id;
 *   use_rccl == 0  ranks that share a GPU (or no RCCL): the library stages the planes through pinned host memory and MPI moves them;
 *   use_rccl == 2  the ranks of ONE node, sharing GPUs or not: peer stores into hipIpc-mapped device windows (fy_comm_create_ipc); MPI carries its bootstrap only */
int fy_mpi_comm_create(const void* mpi_comm, int use_rccl, int device_ordinal, fy_comm** out);
#ifdef __cplusplus
}
#endif
#endif
