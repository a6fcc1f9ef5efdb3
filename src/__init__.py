"""Drop-in module paths of the reference (`src.models.*`, `src.pipelines.*`, `src.cameractrl.*`,
`src.dataset.*`): thin re-exports of the MI355X-native implementation in `humanvid_amd`, so
`scripts/pose2vid.py`-style callers keep their imports."""
