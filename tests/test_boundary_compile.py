"""SURVEY.md §8(b), first sentence: the reference application must compile UNCHANGED against this build's
Controller / ProgramHandler / Logger / FileHandler / Comparator headers and CL/cl.h.

The file is compiled where it lies (/root/reference, never copied), syntax-only, against host/include plus a
test-only declaration stub of the few cv:: names it uses (tests/stubs/opencv2: OpenCV is not installed here).
This is an interface check, not a parity check; it is skipped where /root/reference does not exist (the GPU box)."""
import os
import shutil
import subprocess

import pytest

import __graft_entry__ as entry

REF_APP = "/root/reference/src/RealtimeImageProcessing/RealtimeImageProcessing.cpp"
HOST_INC = os.path.join(entry.PKG_DIR, "host", "include")
STUBS = os.path.join(entry.ROOT, "tests", "stubs")

needs_ref = pytest.mark.skipif(not os.path.exists(REF_APP) or shutil.which("g++") is None,
                               reason="/root/reference (or g++) is not present on this machine")


@needs_ref
def test_reference_application_compiles_unchanged_against_the_host_headers():
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-I", HOST_INC, "-I", STUBS, REF_APP]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-4000:]


@needs_ref
def test_every_member_the_application_calls_is_declared_with_the_reference_signature(tmp_path):
    """Belt and braces for the check above: take the address of every Controller / ProgramHandler / Logger member
    RealtimeImageProcessing.cpp:435-442,351-355,372-376,398-402,423-426 uses, with the exact pointer-to-member
    types of the reference headers (include/Controller.hpp:37-47, include/ProgramHandler.hpp:9-22)."""
    src = tmp_path / "sig.cpp"
    src.write_text(r'''
#include <ProgramHandler.hpp>
#include <FileHandler.hpp>
#include <Comparator.hpp>
using V8 = std::vector<unsigned char>;
using VU = std::vector<cl_ulong>;
void (Controller::*g)(cl_context*, cl_command_queue*, cl_kernel*, VU*, V8*, V8*, cl_int&, cl_int&, Logger&) =
    &Controller::PerformCLImageGrayscaling;
void (Controller::*e)(cl_context*, cl_command_queue*, cl_kernel*, VU*, V8*, V8*, cl_int&, cl_int&, Logger&) =
    &Controller::PerformCLImageEdgeDetection;
void (Controller::*b)(int&, float&, cl_context*, cl_command_queue*, cl_kernel*, VU*, V8*, V8*, cl_int&, cl_int&,
                      Logger&) = &Controller::PerformCLGaussianBlur;
cl_bool (Controller::*gi)() = &Controller::GetImageSupport;
void (ProgramHandler::*io)(Controller&, cl_context*, cl_command_queue*, cl_program*, cl_kernel*, std::string,
                           Logger&) = &ProgramHandler::InitOpenCL;
V8 (ProgramHandler::*po)(Controller&, const cv::Mat&, cl_context*, cl_command_queue*, cl_kernel*, cl_int&, cl_int&,
                         Logger&, std::string) = &ProgramHandler::PerformOpenCL;
V8 (ProgramHandler::*pi)(Controller&, std::string, cl_context*, cl_command_queue*, cl_kernel*, double&, double&,
                         double&, double&, double&, cl_int&, cl_int&, Logger&, std::string) =
    &ProgramHandler::PerformOpenCL;
void (ProgramHandler::*ak)(std::vector<std::string>, std::string) = &ProgramHandler::AddKernels;
void (ProgramHandler::*sd)(int, int) = &ProgramHandler::SetDeviceProperties;
void (ProgramHandler::*il)(Logger&, Logger::LogLevel, bool) = &ProgramHandler::InitLogger;
cv::Mat (Comparator::*cg)(std::string, double&, Logger&) = &Comparator::PerformCPU_Grayscaling;
std::vector<std::string> (FileHandler::*li)(const std::string&) = &FileHandler::LoadImages;
cl_int (*r1)(cl_kernel) = &clReleaseKernel;
cl_int (*r2)(cl_program) = &clReleaseProgram;
cl_int (*r3)(cl_command_queue) = &clReleaseCommandQueue;
cl_int (*r4)(cl_context) = &clReleaseContext;
int main() { ProgramHandler ph(1, false, false, true, true); (void)ph; return 0; }
''')
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-I", HOST_INC, "-I", STUBS, str(src)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-4000:]
