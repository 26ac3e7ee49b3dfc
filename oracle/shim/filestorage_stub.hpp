// oracle/shim/filestorage_stub.hpp — TEST INFRASTRUCTURE.
// cv::FileStorage / cv::FileNode with the members TemplatedVocabulary.h's YAML save / load methods mention, so that the REAL
// Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h compiles against the OpenCV stand-in (those two methods are virtual, hence instantiated, but never
// called here: the vocabulary is read with the reference's own loadFromTextFile).  Every operation aborts.
#pragma once
#include <cstdlib>
#include <string>
namespace cv {
class FileNode {
public:
    enum { NONE = 0, SEQ = 5 };
    FileNode operator[](const std::string&) const { std::abort(); }
    FileNode operator[](const char*) const { std::abort(); }
    FileNode operator[](int) const { std::abort(); }
    int type() const { std::abort(); }
    size_t size() const { std::abort(); }
    operator int() const { std::abort(); }
    operator double() const { std::abort(); }
    operator float() const { std::abort(); }
    operator std::string() const { std::abort(); }
};
class FileStorage {
public:
    enum { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string&, int) { std::abort(); }
    bool isOpened() const { return false; }
    void release() {}
    FileNode operator[](const std::string&) const { std::abort(); }
    FileNode operator[](const char*) const { std::abort(); }
};
template <typename T> inline FileStorage& operator<<(FileStorage& fs, const T&) { std::abort(); return fs; }
}  // namespace cv
