"""CPU: the ros_control plugin tree (adapters/ros_control) must keep compiling against the C ABI / the
C++ adapter.  ROS is not installed here, so the check is `g++ -fsyntax-only` over declaration-only stand-ins of the handful of
ROS / legged_common headers the plugin includes (adapters/ros_control/test_shims); everything from this repository — hunter_hip.h,
hunter_hip.hpp and every call the plugin makes into them — is the real thing."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "adapters/ros_control"


def test_plugin_sources_compile_against_the_abi():
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
                        "-I", str(PKG / "include"), "-I", str(PKG / "test_shims"), "-I", str(ROOT / "include"),
                        str(PKG / "src/HipLeggedController.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_package_files_name_the_plugin_consistently():
    xml = (PKG / "hunter_hip_controllers_plugins.xml").read_text()
    assert 'name="legged/HipLeggedController"' in xml and 'type="legged::HipLeggedController"' in xml
    assert 'base_class_type="controller_interface::ControllerBase"' in xml        # as legged_controllers_plugins.xml:3-8
    cpp = (PKG / "src/HipLeggedController.cpp").read_text()
    assert "PLUGINLIB_EXPORT_CLASS(legged::HipLeggedController, controller_interface::ControllerBase)" in cpp
    cm = (PKG / "CMakeLists.txt").read_text()
    assert "hunter_hip_controllers_plugins.xml" in cm and "find_library(HUNTER_HIP hunter_hip" in cm
    assert "hunter_hip_controllers_plugins.xml" in (PKG / "package.xml").read_text()
