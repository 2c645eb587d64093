"""The C++ drop-in boundary (host/: Controller, ProgramHandler, Logger + the cl* symbols) over the C-ABI.
CPU part: the library builds and exports what RealtimeImageProcessing.cpp links against.
GPU part: host_app drives it the way the reference app does and its outputs are checked with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as entry

LIBDIR = os.path.join(entry.PKG_DIR, "lib")
HOST_SO = os.path.join(LIBDIR, "libmi355_host.so")
HOST_APP = os.path.join(LIBDIR, "host_app")
HOST_ERRORS = os.path.join(LIBDIR, "host_errors")


@pytest.fixture(scope="module")
def host_built(pkg):
    if not (os.path.exists(HOST_SO) and os.path.exists(HOST_APP) and os.path.exists(HOST_ERRORS)):
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(entry.PKG_DIR, "host")], check=True)
    return HOST_SO


def test_host_library_exports_the_reference_surface(host_built):
    syms = subprocess.run(["nm", "-D", "--defined-only", "-C", host_built], check=True, capture_output=True,
                          text=True).stdout
    # C symbols the unchanged app / ProgramHandler reference directly
    for c_sym in ("clGetDeviceInfo", "clGetPlatformInfo", "clReleaseKernel", "clReleaseProgram",
                  "clReleaseCommandQueue", "clReleaseContext", "clReleaseMemObject", "clReleaseSampler"):
        assert (" T " + c_sym) in syms, c_sym
    # the C++ members with the reference's exact signatures (include/Controller.hpp:37-47 etc.)
    for cxx in (
        "Controller::PerformCLImageGrayscaling(_cl_context**, _cl_command_queue**, _cl_kernel**, "
        "std::vector<unsigned long, std::allocator<unsigned long> >*, "
        "std::vector<unsigned char, std::allocator<unsigned char> >*, "
        "std::vector<unsigned char, std::allocator<unsigned char> >*, int&, int&, Logger&)",
        "Controller::PerformCLGaussianBlur(int&, float&, _cl_context**, _cl_command_queue**, _cl_kernel**",
        "Controller::PerformCLImageEdgeDetection(_cl_context**",
        "Controller::CreateProgram(_cl_context*, _cl_device_id*, char const*)",
        "Controller::CreateKernel(_cl_program*, char const*)",
        "Controller::Cleanup(_cl_context*, _cl_command_queue*, _cl_program*, _cl_kernel*, _cl_sampler*, _cl_mem**, int)",
        "ProgramHandler::ProgramHandler(int, bool, bool, bool, bool, int, float)",
        "ProgramHandler::InitOpenCL(Controller&, _cl_context**, _cl_command_queue**, _cl_program**, _cl_kernel**, "
        "std::__cxx11::basic_string",
        "ProgramHandler::PerformOpenCL(Controller&, cv::Mat const&, _cl_context**",
        "Logger::getInstance()",
        "Logger::log(std::__cxx11::basic_string",
        "FileHandler::WriteResultsToCSV(",
    ):
        assert cxx in syms, cxx


def test_host_library_contains_no_cpu_filter(host_built):
    """No CPU fallback in the product: the host layer has no pixel loop, it only forwards."""
    src_dir = os.path.join(entry.PKG_DIR, "host", "src")
    text = "".join(open(os.path.join(src_dir, f)).read() for f in os.listdir(src_dir))
    assert "0.299" not in text and "0.587" not in text and "sqrt(" not in text
    assert "mi355_gray_rgba8" in text and "mi355_gauss_rgba8" in text and "mi355_sobel_rgba8" in text


@pytest.mark.gpu
def test_host_app_matches_oracle(host_built, oracle, tmp_path, fixture_rgb):
    w, h = 500, 131
    frame = oracle.synth_rgba(w, h, 1, first_frame=5, mode=1)[0]
    raw = tmp_path / "in.rgba"
    raw.write_bytes(frame.tobytes())
    ppm = tmp_path / "img.ppm"
    crop = np.ascontiguousarray(fixture_rgb[:96, :128])
    ppm.write_bytes(b"P6\n# fixture crop\n128 96\n255\n" + crop.tobytes())
    prefix = str(tmp_path / "out")
    # the reference's ProgramHandler opens RealtimeImageProcessing.log in the working directory (ProgramHandler.cpp:14)
    # default: CL_DEVICE_IMAGE_SUPPORT = CL_FALSE — a BYPASS_IMAGE_SUPPORT = false application stays on the
    # buffer kernels (host_app checks that its grayscale equals the buffer-mode one and stops there)
    env = {k: v for k, v in os.environ.items() if k != "MI355_CL_IMAGE_SUPPORT"}
    run = subprocess.run([HOST_APP, str(raw), str(w), str(h), prefix, str(ppm)], capture_output=True, text=True,
                         timeout=300, cwd=str(tmp_path), env=env)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "host_app ok (image support not opted in)" in run.stdout
    assert "Device does not support images. Using buffers instead of image2D structures." in run.stdout
    # the explicit opt-in: the whole run again, image2d_t semantics for the BYPASS = false ProgramHandler
    run = subprocess.run([HOST_APP, str(raw), str(w), str(h), prefix, str(ppm)], capture_output=True, text=True,
                         timeout=300, cwd=str(tmp_path), env=dict(env, MI355_CL_IMAGE_SUPPORT="1"))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "host_app ok\n" in run.stdout and "Device supports images." in run.stdout
    # the reference's bootstrap chatter is preserved (Controller.cpp:25,35,62,111,127,177,189)
    for line in ("Number of platforms: 1", "Successfully created a context", "Successfully created CommandQueue",
                 "Successfully created a program", "Successfully created the gaussian_blur kernel",
                 "Bypass image support is True. Using buffers instead of image2D structures."):
        assert line in run.stdout, line

    def load(ext, shape):
        return np.fromfile(prefix + ext, dtype=np.uint8).reshape(shape)

    assert np.array_equal(load(".gray", (h, w, 4)), oracle.gray_rgba(frame))
    assert np.array_equal(load(".edge", (h, w)), oracle.sobel_rgba(frame))
    d = np.abs(load(".gauss", (h, w, 4)).astype(int) - oracle.gauss_rgba(frame, 5, 1.5).astype(int))
    assert d.max() <= 1
    d = np.abs(load(".gauss17", (h, w, 4)).astype(int) - oracle.gauss_rgba(frame, 17, 6.0).astype(int))
    assert d.max() <= 1
    assert np.array_equal(np.fromfile(prefix + ".weights", dtype=np.float32).view(np.uint32),
                          oracle.gauss_weights(5, 1.5).reshape(-1).view(np.uint32))
    prof = np.fromfile(prefix + ".prof", dtype=np.uint64)
    assert prof.size == 12 and (np.diff(prof[:6].astype(np.int64)) >= 0).all() and prof[6] >= prof[5]
    rgba = np.dstack([crop, np.full(crop.shape[:2], 255, np.uint8)])
    assert np.array_equal(load(".ppm_gray", (96, 128, 4)), oracle.gray_rgba(rgba))
    # ProgramHandler::PerformOpenCLBatch (MI355X extension): five frames over a two-member group; host_app itself
    # checked EDGE / GAUSSIAN against the per-frame calls, the fused pipeline is checked here against the chained oracle
    assert np.array_equal(load(".batch_pipe", (h, w)), oracle.pipeline_rgba(frame, 5, 1.5))
    # BYPASS_IMAGE_SUPPORT = false: image2d_t semantics and output shapes ("w*h gray bytes then zeros")
    ig = load(".img_gray", (h * w * 4,))
    assert np.array_equal(ig[: h * w].reshape(h, w), oracle.image2d_gray(frame)) and not ig[h * w:].any()
    assert np.array_equal(load(".img_edge", (h, w)), oracle.image2d_sobel(frame))
    assert np.array_equal(load(".img_gauss", (h, w, 4)), oracle.image2d_gauss(frame, 5, 1.5))
    assert np.array_equal(np.fromfile(prefix + ".img_weights", dtype=np.float32).view(np.uint32),
                          oracle.gauss_weights_image2d(5, 1.5).reshape(-1).view(np.uint32))


def _run_err(what):
    import tempfile
    with tempfile.TemporaryDirectory() as d:   # log files land there, not in the repository
        return subprocess.run([HOST_ERRORS, what], capture_output=True, text=True, timeout=120, cwd=d)


def test_logger_throws_like_the_reference(host_built):
    r = _run_err("log_throw")
    assert r.returncode == 0 and "caught: Failed to open log file: /nonexistent-dir/x/y.log" in r.stdout


@pytest.mark.gpu
def test_error_conventions_match_the_reference(host_built):
    """Exit codes and messages of the reference (SURVEY.md §8b "Error conventions")."""
    r = _run_err("unknown_method")
    assert r.returncode == 1 and "Unrecognised method" in r.stderr              # ProgramHandler.cpp:75-78
    r = _run_err("bad_kernel")
    assert r.returncode == 1 and "Error: clCreateKernel (-46)" in r.stderr      # Controller.cpp:5-11 (EXIT_FAILURE)
    r = _run_err("bad_program")
    assert r.returncode == 0 and "program is NULL" in r.stdout and "Failed to open file for reading: foo.cl" in r.stderr
    r = _run_err("short_input")
    assert r.returncode == 0 and "output untouched" in r.stdout and "[ERROR] Failed to write cl_mem" in r.stdout
    r = _run_err("even_kernel")
    assert r.returncode == 0 and "0 events" in r.stdout and "[ERROR] Failed when executing kernel: bad argument" in r.stdout


REF_CSV_HEADER = ("Timestamp, Image, Resolution, Num_Iterations, avg_CPU_Time_ms, avg_OpenCL_Time_ms, "
                  "avg_OpenCL_kernel_ms, avg_OpenCL_kernel_write_ms, avg_OpenCL_kernel_read_ms, "
                  "avg_OpenCL_kernel_operation_ms, Error_MAE")   # RT/src/FileHandler.cpp:28, byte for byte


def test_csv_writer_and_directory_scan_like_the_reference(host_built, tmp_path):
    """FileHandler::WriteResultsToCSV writes the reference's file (header byte for byte, ", "-separated rows through
    operator<< of double: RT/src/FileHandler.cpp:25-34); LoadImages scans a directory for images (:5-14).  No GPU."""
    out = tmp_path / "results.csv"
    r = subprocess.run([HOST_ERRORS, "csv", str(out)], capture_output=True, text=True, timeout=60, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    lines = out.read_text().split("\n")
    assert lines[0] == REF_CSV_HEADER and lines[-1] == ""
    assert lines[1] == "2026-10-04 12:00:00, a.jpg, 640x512, 100, 0.30868, 1.512, 0.05, 0.7, 0.4, 1.15, 0.000410156"
    assert lines[2] == "2026-10-04 12:00:01, b.jpg, 75x75, 3, 1, 2, 0.5, 0.25, 0.125, 0.875, 0"
    d = tmp_path / "imgs"
    d.mkdir()
    for name in ("b.png", "a.jpg", "notes.txt", "c.ppm"):
        (d / name).write_bytes(b"x")
    r = subprocess.run([HOST_ERRORS, "load_images", str(d)], capture_output=True, text=True, timeout=60, cwd=str(tmp_path))
    assert r.returncode == 0
    assert [os.path.basename(p) for p in r.stdout.split()] == ["a.jpg", "b.png", "c.ppm"]


def test_host_layer_is_clean_under_address_and_undefined_sanitizers(pkg, tmp_path):
    """host code is built with -fsanitize=address,undefined for tests.  The host sources + the error
    scenarios that need no device, compiled with both sanitizers, run clean (leak detection off: the HIP runtime the
    C-ABI library links keeps process-lifetime allocations)."""
    host = os.path.join(entry.PKG_DIR, "host")
    srcs = [os.path.join(host, "src", f) for f in sorted(os.listdir(os.path.join(host, "src"))) if f.endswith(".cpp")]
    exe = tmp_path / "host_errors_asan"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-I", os.path.join(host, "include"), "-I", os.path.join(entry.ROOT, "include"),
           os.path.join(host, "tests", "host_errors.cpp")] + srcs + \
          ["-L", LIBDIR, "-lmi355_imgfilter", "-Wl,-rpath," + LIBDIR, "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for args in (["log_throw"], ["csv", str(tmp_path / "r.csv")], ["load_images", str(tmp_path)]):
        r = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=120, env=env, cwd=str(tmp_path))
        assert r.returncode == 0, (args, r.stderr[-2000:])
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
