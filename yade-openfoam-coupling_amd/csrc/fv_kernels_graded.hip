// The FV kernels for a GRADED (rectilinear) single block: fv_kernels.hip compiled with the general geometry model -- per-axis cell sizes,
// linear-interpolation weights != 1/2, |Sf| / |d| per face (blockMesh simpleGrading; icoFoamYade/createFields.H:15-162 and
// pimpleFoamYade/createFields.H:32-261 take any fvMesh) -- into namespace fy::gr.  The uniform block keeps its own build of the same
// source (namespace fy), whose kernels work with the constants dx, Af, V.
#define FY_FVK_GRADED 1
#include "fv_kernels.hip"
