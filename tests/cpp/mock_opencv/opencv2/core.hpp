// tests/cpp/mock_opencv/opencv2/core.hpp -- a MOCK of the handful of OpenCV core names that include/sbm_stereobm.hpp's
// cv::InputArray / cv::OutputArray overload and tests/cpp/callsite_main.cpp use.  Test infrastructure only.
//
// Written from scratch from the list of names that overload touches (cv::Mat, cv::Mat_<T>, cv::Rect, cv::Size,
// cv::InputArray / cv::OutputArray with getMat / create / fixedType / type / size, Mat::convertTo, CV_Error,
// cv::Exception); it is not derived from OpenCV's headers and reproduces none of OpenCV's arithmetic.  What it is for:
// this image (and the GPU box) has no OpenCV, so the overload the maintainer's one-line diff at
// src/slam/src/core/main.cpp:201-215 relies on had never been through a compiler.  Building against this mock proves that
// text compiles and runs -- the call shape, the CV_16SC1 / fixed-CV_32F destination rule, the error -> cv::Exception
// mapping.  It pins NOTHING about cv::StereoBM's results; where real OpenCV headers exist the test uses those instead.
#ifndef SBM_MOCK_OPENCV_CORE_HPP_
#define SBM_MOCK_OPENCV_CORE_HPP_

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#define SBM_MOCK_OPENCV 1

// type codes: depth in the low 3 bits, (channels - 1) above them -- only the three single-channel types below are ever used
#define CV_8U 0
#define CV_16S 3
#define CV_32F 5
#define CV_8UC1 CV_8U
#define CV_16SC1 CV_16S
#define CV_32FC1 CV_32F

namespace cv {

namespace Error {
enum Code { StsError = -2, StsOutOfRange = -211, StsUnmatchedSizes = -209, StsUnsupportedFormat = -210 };
}

class Exception : public std::exception {
 public:
  Exception(int c, const std::string& m) : code(c), err(m) {}
  const char* what() const noexcept override { return err.c_str(); }
  int code;
  std::string err;
};

struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
  bool operator==(const Size& o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size& o) const { return !(*this == o); }
};

struct Rect {
  int x, y, width, height;
  Rect() : x(0), y(0), width(0), height(0) {}
  Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

class _OutputArray;

// dense row-major single-channel matrix: either a view of caller memory or the owner of a shared buffer
class Mat {
 public:
  struct Step {   // converts like cv::MatStep: bytes per row
    size_t v = 0;
    operator size_t() const { return v; }
  };
  int rows = 0, cols = 0;
  Step step;
  Mat() {}
  Mat(int r, int c, int type, void* external) : rows(r), cols(c), type_(type), data_(static_cast<unsigned char*>(external)) { step.v = (size_t)c * esz(type); }
  static size_t esz(int type) { return type == CV_8U ? 1 : (type == CV_16S ? 2 : 4); }
  int type() const { return type_; }
  Size size() const { return Size(cols, rows); }
  bool empty() const { return data_ == nullptr; }
  void create(Size s, int type) {
    if (data_ && rows == s.height && cols == s.width && type_ == type) return;
    own_ = std::make_shared<std::vector<unsigned char>>((size_t)s.width * s.height * esz(type));
    data_ = own_->data(); rows = s.height; cols = s.width; type_ = type; step.v = (size_t)s.width * esz(type);
  }
  void create(int r, int c, int type) { create(Size(c, r), type); }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data_ + (size_t)r * step.v); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data_ + (size_t)r * step.v); }
  // only the conversion the adaptor needs: CV_16S -> CV_32F with a scale factor
  inline void convertTo(const _OutputArray& dst, int rtype, double alpha = 1.0) const;

 protected:
  int type_ = CV_8U;
  unsigned char* data_ = nullptr;
  std::shared_ptr<std::vector<unsigned char>> own_;
};

// matrix with a compile-time element type: as an output argument its type is FIXED
template <class T> struct MatTypeOf;
template <> struct MatTypeOf<unsigned char> { enum { value = CV_8U }; };
template <> struct MatTypeOf<short> { enum { value = CV_16S }; };
template <> struct MatTypeOf<float> { enum { value = CV_32F }; };
template <class T>
class Mat_ : public Mat {
 public:
  Mat_() { type_ = MatTypeOf<T>::value; }
  Mat_(int r, int c) { type_ = MatTypeOf<T>::value; create(r, c, MatTypeOf<T>::value); }
};

class _InputArray {
 public:
  _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}   // NOLINT: implicit, like cv::InputArray
  Size size() const { return m_->size(); }
  int type() const { return m_->type(); }
  Mat getMat() const { return *m_; }

 protected:
  Mat* m_;
};

class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m) : _InputArray(m), fixed_(false) {}   // NOLINT
  template <class T> _OutputArray(Mat_<T>& m) : _InputArray(m), fixed_(true) {}   // NOLINT
  bool fixedType() const { return fixed_; }
  void create(Size s, int type) const {
    if (fixed_ && type != m_->type()) throw Exception(Error::StsUnsupportedFormat, "mock: a fixed-type destination cannot change its type");
    m_->create(s, type);
  }

 private:
  bool fixed_;
};

typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

inline void Mat::convertTo(const _OutputArray& dst, int rtype, double alpha) const {
  if (type_ != CV_16S || rtype != CV_32F) throw Exception(Error::StsUnsupportedFormat, "mock: only CV_16S -> CV_32F is implemented");
  dst.create(size(), CV_32F);
  Mat out = dst.getMat();
  for (int y = 0; y < rows; y++) {
    const short* s = ptr<short>(y);
    float* d = out.ptr<float>(y);
    for (int x = 0; x < cols; x++) d[x] = (float)(s[x] * alpha);
  }
}

}  // namespace cv

#define CV_Error(code, msg) throw cv::Exception((int)(code), std::string(msg))

#endif
