// simple_main.cpp -- drop-in command line: ./simple <A.fasta> <B.fasta> <p>
// Same contract as the reference's main (simple.cpp:297-313): inputs are read from
// "sequences/<name>" relative to the working directory, p is parsed with std::stoi, the five
// result lines go to stdout, exit status 0 even when a file is missing (the reader then yields
// an empty sequence).  The alignment itself runs on the GPU through the C ABI of include/psa.h;
// there is no CPU path, so without a usable device the program reports the error and exits 3.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "../../include/psa.h"

int main(int argc, char* argv[]) {
    if (argc != 4) {
        std::cerr << "Usage: " << argv[0] << " <sequence1.fasta> <sequence2.fasta> <p>\n";
        return 1;
    }
    const std::string folder = "sequences/";
    const std::string file_a = folder + argv[1];
    const std::string file_b = folder + argv[2];
    const int p = std::stoi(argv[3]);  // non-numeric input aborts, as in the reference

    int64_t n = 0, m = 0;
    uint8_t* a = psa_read_fasta(file_a.c_str(), &n);
    uint8_t* b = psa_read_fasta(file_b.c_str(), &m);

    int32_t score = 0;
    int32_t *ai = nullptr, *bj = nullptr;
    int64_t len = 0;
    // PSA_GPUS=N (N > 1): run()'s workers on N GPUs of this process -- B cut into N column stripes, one per GPU
    // (psa_align_multi); the output does not depend on it, as the reference's does not depend on p.
    const char* gpus_env = std::getenv("PSA_GPUS");
    const int gpus = gpus_env ? std::atoi(gpus_env) : 0;
    const int rc = gpus > 1 ? psa_align_multi(a, n, b, m, nullptr, gpus, &score, &ai, &bj, &len)
                            : psa_align(a, n, b, m, p, &score, &ai, &bj, &len);
    if (rc != PSA_OK) {
        std::cerr << "simple: " << psa_last_error() << " (code " << rc << ")\n";
        return 3;
    }
    int64_t out_len = 0;
    char* text = psa_format_result(score, a, b, ai, bj, len, &out_len);
    fwrite(text, 1, (size_t)out_len, stdout);
    fflush(stdout);
    psa_free(text); psa_free(ai); psa_free(bj);

    // The reference's timing sweep (commented out in simple.cpp:319-335): p = 1..32, one line per p on
    // stdout and "threads,time_microseconds" rows in a CSV.  Off unless PSA_TIMINGS_CSV names the file,
    // so that the default output stays the reference's five lines.
    if (const char* csv_path = std::getenv("PSA_TIMINGS_CSV")) {
        std::ofstream csv(csv_path);
        csv << "threads,time_microseconds\n";
        for (int q = 1; q <= 32; ++q) {
            const auto start = std::chrono::high_resolution_clock::now();
            int32_t sc = 0, *xi = nullptr, *xj = nullptr;
            int64_t xl = 0;
            const int rq = psa_align(a, n, b, m, q, &sc, &xi, &xj, &xl);
            const auto end = std::chrono::high_resolution_clock::now();
            if (rq != PSA_OK) {
                std::cerr << "simple: " << psa_last_error() << " (code " << rq << ")\n";
                return 3;
            }
            psa_free(xi); psa_free(xj);
            const auto us = std::chrono::duration_cast<std::chrono::microseconds>(end - start).count();
            std::cout << "p=" << q << ", time=" << us << " \xC2\xB5s, score=" << sc << std::endl;
            csv << q << "," << us << "\n";
        }
        csv.close();
        std::cout << "Timing results written to " << csv_path << "\n";
    }
    psa_free(a); psa_free(b);
    return 0;
}
