// Pure host arithmetic of kernel SPLIT's virtual rows (kernels_csr_split.hip: csr_split_build, mode 2): no HIP, no state, so
// that tests/test_abi_and_host.py can compile it with g++ (tests/split_rows_check.cpp).
//
// A long row of `len` entries becomes V = ceil(len / per) virtual rows; entry k of the row (k = 0 .. len - 1, in column order)
// goes to virtual row k mod V, position k / V: neighbours in x land in different virtual rows, every virtual row keeps its
// entries in column order, and the virtual rows differ in length by at most one.
#pragma once
#include <cstdint>
#include <vector>

namespace spmv
{
inline int32_t split_virtual_rows(int64_t len, int per) { return (int32_t)((len + per - 1) / per); }

// entries of virtual row v (0 .. V - 1) of a long row of `len` entries
inline int64_t split_virtual_len(int64_t len, int32_t V, int32_t v) { return len / V + (v < len % V ? 1 : 0); }

// where entry k of the long row lands: (virtual row, position inside it)
inline void split_deal(int64_t k, int32_t V, int32_t* vrow, int64_t* pos)
{
    *vrow = (int32_t)(k % V);
    *pos  = k / V;
}

// row_ptr of the virtual rows' matrix for long rows of the given lengths; lbase[i] = first virtual row of long row i
inline void split_virtual_row_ptr(const std::vector<int64_t>& lens, int per, std::vector<int32_t>* lv, std::vector<int32_t>* lbase, std::vector<int32_t>* vptr)
{
    lv->assign(lens.size(), 0);
    lbase->assign(lens.size(), 0);
    int64_t nv = 0;
    for (size_t i = 0; i < lens.size(); ++i)
    {
        (*lv)[i]    = split_virtual_rows(lens[i], per);
        (*lbase)[i] = (int32_t)nv;
        nv += (*lv)[i];
    }
    vptr->assign((size_t)nv + 1, 0);
    int64_t at = 0;
    for (size_t i = 0; i < lens.size(); ++i)
        for (int32_t v = 0; v < (*lv)[i]; ++v)
        {
            (*vptr)[(size_t)(*lbase)[i] + (size_t)v] = (int32_t)at;
            at += split_virtual_len(lens[i], (*lv)[i], v);
        }
    (*vptr)[(size_t)nv] = (int32_t)at;
}
}  // namespace spmv
