{
  "targets": [
    {
      "target_name": "tidalwave",
      "sources": [ "addon.cc", "twhost.cpp", "jpeg_gray.cpp", "tw_inflate.cpp" ],
      "cflags_cc": [ "-std=c++17", "-O2" ],
      "include_dirs": [ "../../include" ],
      "libraries": [ "-L<(module_root_dir)/..", "-ltwflow", "-Wl,-rpath,<(module_root_dir)/.." ]
    }
  ]
}
