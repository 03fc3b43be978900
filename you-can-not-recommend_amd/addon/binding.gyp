{
  "targets": [
    {
      "target_name": "ycnr_als",
      "sources": ["ycnr_als_napi.cc"],
      "include_dirs": ["../../include"],
      "libraries": ["-L<(module_root_dir)/../csrc", "-lycnr_als", "-Wl,-rpath,<(module_root_dir)/../csrc"],
      "cflags_cc": ["-std=c++14", "-O2"]
    }
  ]
}
