// rvtests_amd — the one accessor of the in-tree binding that needs Eigen's headers (third/eigen, downloaded by the
// rvtests build: third/Makefile:22-26): the float storage of an EigenMatrix (regression/EigenMatrix.h:9-12), column-major,
// exactly what rvt_set_kinship takes.  Compiled inside the rvtests tree only.
#include "regression/EigenMatrix.h"

namespace rvt_intree {
const float* eigenMatrixData(const EigenMatrix* m) { return m ? m->mat.data() : 0; }
}  // namespace rvt_intree
