// interface shim (tests/faiss_shim/README.md): nsg::Graph<node_t> as altid_impl.h:29-67 subclasses it
#pragma once
#include <cstddef>
namespace faiss {
namespace nsg {
template <class node_t>
struct Graph {
    node_t* data;
    int K;
    int N;
    bool own_fields;
    Graph(node_t* data_, int N_, int K_) : data(data_), K(K_), N(N_), own_fields(false) {}
    virtual ~Graph() {}
    inline node_t at(int i, int j) const { return data[i * K + j]; }
    virtual size_t get_neighbors(int i, node_t* neighbors) const {
        for (int j = 0; j < K; j++) {
            if (data[i * K + j] < 0) return j;
            neighbors[j] = data[i * K + j];
        }
        return K;
    }
};
}  // namespace nsg
}  // namespace faiss
