// spmm_test -- ./spmm_test method(1:partition-B, 2:partition-A) A_path B_width alpha beta gpus
// Same positional command line as the reference's spmm_test.cu:97-108; gpus = 0 runs the host path only.
#include "harness.h"

int main(int argc, char *argv[])
{
    if (argc != 7) {
        cerr << "./spmm_test method(1:partition-B, 2:partition-A) A_path B_width alpha beta gpus" << endl;
        return 1;
    }
    const int method = atoi(argv[1]);
    if (method != 1 && method != 2) {
        cerr << "Method can be only 1 or 2." << endl;
        return 1;
    }
    const bool ok = harness::spmm(method, argv[2], (int)atof(argv[3]), atof(argv[4]), atof(argv[5]), (unsigned)atoi(argv[6]));
    return ok ? 0 : 2;
}
