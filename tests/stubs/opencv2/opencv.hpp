// TEST-ONLY declaration stub — not OpenCV and not shipped.  It declares exactly the cv:: names that the
// reference application src/RealtimeImageProcessing/RealtimeImageProcessing.cpp and the headers it includes use,
// so that tests/test_boundary_compile.py can run `g++ -fsyntax-only` on that file IN PLACE against
// host/include ("must compile unchanged").  An interface check only: nothing here is
// implemented, linked, or used for parity.
#ifndef MI355_TEST_OPENCV_STUB_HPP
#define MI355_TEST_OPENCV_STUB_HPP

#include <cstddef>
#include <iostream>
#include <string>
#include <vector>

typedef unsigned char uchar;

#define CV_8UC1 0
#define CV_8UC3 16
#define CV_8UC4 24

namespace cv {

struct Size {
    Size(int w, int h);
    int width, height;
};
struct Point {
    Point(int x, int y);
    int x, y;
};
struct Scalar {
    Scalar(double v0 = 0, double v1 = 0, double v2 = 0, double v3 = 0);
};

class Mat
{
public:
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, void* data, size_t step = 0);
    int rows, cols;
    unsigned char* data;
    bool empty() const;
    size_t total() const;
    int channels() const;
    int type() const;
    Mat clone() const;
    template <typename T> T* ptr(int row = 0);
    template <typename T> const T* ptr(int row = 0) const;
};

enum ColorConversionCodes { COLOR_BGR2RGBA = 2, COLOR_RGBA2BGR = 3, COLOR_BGR2GRAY = 6 };
enum VideoCaptureAPIs { CAP_ANY = 0, CAP_GSTREAMER = 1800 };
enum VideoCaptureProperties { CAP_PROP_FPS = 5 };
enum HersheyFonts { FONT_HERSHEY_SIMPLEX = 0 };
enum ImreadModes { IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1 };

void cvtColor(const Mat& src, Mat& dst, int code);
void resize(const Mat& src, Mat& dst, Size dsize);
void imshow(const std::string& winname, const Mat& mat);
int waitKey(int delay = 0);
void destroyAllWindows();
void putText(Mat& img, const std::string& text, Point org, int fontFace, double fontScale, Scalar color,
             int thickness = 1);
Mat imread(const std::string& filename, int flags = IMREAD_COLOR);
bool imwrite(const std::string& filename, const Mat& img);

class VideoCapture
{
public:
    VideoCapture();
    explicit VideoCapture(int index, int apiPreference = CAP_ANY);
    explicit VideoCapture(const std::string& filename, int apiPreference = CAP_ANY);
    bool isOpened() const;
    double get(int propId) const;
    void release();
    VideoCapture& operator>>(Mat& image);
    bool read(Mat& image);
};

}  // namespace cv

#endif  // MI355_TEST_OPENCV_STUB_HPP
