// interface shim (tests/faiss_shim/README.md): FaissException + the two assertion macros the adapter uses
#pragma once
#include <stdexcept>
#include <string>
namespace faiss {
struct FaissException : std::runtime_error {
    explicit FaissException(const std::string& m) : std::runtime_error(m) {}
};
}  // namespace faiss
#define FAISS_THROW_IF_NOT_MSG(X, MSG)                                                     \
    do {                                                                                   \
        if (!(X)) throw faiss::FaissException(std::string("Error: '" #X "' failed: ") + (MSG)); \
    } while (0)
#define FAISS_THROW_IF_NOT(X)                                                              \
    do {                                                                                   \
        if (!(X)) throw faiss::FaissException("Error: '" #X "' failed");                    \
    } while (0)
