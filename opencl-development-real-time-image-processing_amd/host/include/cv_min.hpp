// cv_min.hpp — used ONLY when OpenCV is not installed (this build container, the GPU box): the two
// things ProgramHandler's signatures need from cv:: — a dense 8-bit matrix view (rows, cols, data,
// total(), channels()) — and nothing else.  With OpenCV present <opencv2/opencv.hpp> is used instead and
// this file is not included.  It exists so the host layer can be built and tested without OpenCV; it is
// not an OpenCV replacement.
#ifndef MI355_CV_MIN_HPP
#define MI355_CV_MIN_HPP

#include <cstddef>
#include <memory>
#include <vector>

#define MI355_NO_OPENCV 1

namespace cv {

class Mat
{
public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;

    Mat() = default;
    Mat(int r, int c, int nchannels) : rows(r), cols(c), m_ch(nchannels),
        m_own(std::make_shared<std::vector<unsigned char>>((size_t)r * c * nchannels))
    {
        data = m_own->data();
    }
    // view over caller-owned pixels (like cv::Mat(rows, cols, type, ptr))
    Mat(int r, int c, int nchannels, unsigned char* ptr) : rows(r), cols(c), data(ptr), m_ch(nchannels) {}

    size_t total() const { return (size_t)rows * cols; }
    int channels() const { return m_ch; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }

private:
    int m_ch = 0;
    std::shared_ptr<std::vector<unsigned char>> m_own;
};

}  // namespace cv

#endif  // MI355_CV_MIN_HPP
