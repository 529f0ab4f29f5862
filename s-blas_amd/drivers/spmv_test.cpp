// spmv_test -- ./spmv_test A_path alpha beta gpus   (reference spmv_test.cu:45-55; gpus = 0: host path only)
#include "harness.h"

int main(int argc, char *argv[])
{
    if (argc != 5) {
        cerr << "./spmv_test A_path alpha beta gpus" << endl;
        return 1;
    }
    return harness::spmv(argv[1], atof(argv[2]), atof(argv[3]), (unsigned)atoi(argv[4])) ? 0 : 2;
}
