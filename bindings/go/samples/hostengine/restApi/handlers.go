package main

import (
	"encoding/json"
	"fmt"
	"log"
	"net/http"
	"strconv"
	"strings"
	"time"

	he "github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/hostengine"
)

// api routes by hand: the path grammar is small and fixed, so no router package.
type api struct {
	uuids map[string]uint // uuid -> GPU id, built once at start (restApi/handlers/byUuids.go:13-29 of the reference)
	count uint
	pids  he.GroupHandle
}

func newAPI() (*api, error) {
	a := &api{uuids: map[string]uint{}}
	n, err := he.GetAllDeviceCount()
	if err != nil {
		return nil, err
	}
	a.count = n
	for i := uint(0); i < n; i++ {
		d, err := he.GetDeviceInfo(i)
		if err != nil {
			return nil, err
		}
		a.uuids[d.UUID] = i
	}
	if a.pids, err = he.WatchPidFields(); err != nil {
		return nil, err
	}
	return a, nil
}

type httpError struct {
	code int
	msg  string
}

func (e *httpError) Error() string { return e.msg }

// gpuFromPath resolves "id/<n>" or "uuid/<u>": 400 for an unparsable id, 404 for an unknown GPU.
func (a *api) gpuFromPath(kind, value string) (uint, error) {
	switch kind {
	case "id":
		n, err := strconv.ParseUint(value, 10, 32)
		if err != nil {
			return 0, &httpError{http.StatusBadRequest, fmt.Sprintf("bad GPU id %q", value)}
		}
		if uint(n) >= a.count {
			return 0, &httpError{http.StatusNotFound, fmt.Sprintf("GPU %d not found", n)}
		}
		return uint(n), nil
	case "uuid":
		id, ok := a.uuids[value]
		if !ok {
			return 0, &httpError{http.StatusNotFound, fmt.Sprintf("GPU %s not found", value)}
		}
		return id, nil
	}
	return 0, &httpError{http.StatusNotFound, "not found"}
}

func (a *api) ServeHTTP(w http.ResponseWriter, r *http.Request) {
	path := strings.Trim(r.URL.Path, "/")
	parts := strings.Split(path, "/")
	if len(parts) < 2 || (parts[0] != "dcgm" && parts[0] != "gpumon") || r.Method != http.MethodGet {
		http.NotFound(w, r)
		return
	}
	parts = parts[1:]
	asJSON := parts[len(parts)-1] == "json"
	if asJSON {
		parts = parts[:len(parts)-1]
	}
	var body interface{}
	var err error
	switch {
	case len(parts) == 4 && parts[0] == "device" && parts[1] == "info":
		var gpu uint
		if gpu, err = a.gpuFromPath(parts[2], parts[3]); err == nil {
			body, err = he.GetDeviceInfo(gpu)
		}
	case len(parts) == 4 && parts[0] == "device" && parts[1] == "status":
		var gpu uint
		if gpu, err = a.gpuFromPath(parts[2], parts[3]); err == nil {
			body, err = he.GetDeviceStatus(gpu)
		}
	case len(parts) == 3 && parts[0] == "health":
		var gpu uint
		if gpu, err = a.gpuFromPath(parts[1], parts[2]); err == nil {
			body, err = he.HealthCheckByGpuId(gpu)
		}
	case len(parts) == 4 && parts[0] == "process" && parts[1] == "info" && parts[2] == "pid":
		pid, perr := strconv.ParseUint(parts[3], 10, 32)
		if perr != nil {
			err = &httpError{http.StatusBadRequest, fmt.Sprintf("bad pid %q", parts[3])}
			break
		}
		// Accounting is already running (started in newAPI), so there is no 3 s warm-up sleep per request
		// as in the reference (restApi/handlers/dcgm.go:129); one forced sample makes the answer current.
		_ = he.UpdateAllFields()
		body, err = he.GetProcessInfo(a.pids, uint(pid))
		if he.IsStatus(err, he.StatusNotFound) || he.IsStatus(err, he.StatusNoData) {
			err = &httpError{http.StatusNotFound, fmt.Sprintf("no GPU activity recorded for pid %d", pid)}
		}
	case len(parts) == 1 && parts[0] == "status":
		body, err = he.Introspect()
	default:
		http.NotFound(w, r)
		return
	}
	if err != nil {
		code := http.StatusInternalServerError
		if he.IsStatus(err, he.StatusGpuNotSupported) {
			code = http.StatusInternalServerError // unsupported GPU: same answer as the reference (utils.go:131-147)
		}
		if h, ok := err.(*httpError); ok {
			code = h.code
		}
		log.Printf("error: %v%v: %v", r.Host, r.URL, err)
		http.Error(w, err.Error(), code)
		return
	}
	if asJSON {
		w.Header().Set("Content-Type", "application/json")
		enc := json.NewEncoder(w)
		enc.SetIndent("", "  ")
		_ = enc.Encode(body)
		return
	}
	w.Header().Set("Content-Type", "text/plain; charset=utf-8")
	fmt.Fprintf(w, "%s\n", render(body))
}

// render prints a struct as aligned "Name : value" lines, following pointers; nil prints N/A.
func render(v interface{}) string {
	b, _ := json.Marshal(v)
	var generic interface{}
	_ = json.Unmarshal(b, &generic)
	var sb strings.Builder
	var walk func(prefix string, x interface{})
	walk = func(prefix string, x interface{}) {
		switch t := x.(type) {
		case map[string]interface{}:
			for _, k := range sortedKeys(t) {
				walk(strings.TrimPrefix(prefix+" "+k, " "), t[k])
			}
		case []interface{}:
			if len(t) == 0 {
				fmt.Fprintf(&sb, "%-32s: None\n", prefix)
			}
			for i, e := range t {
				walk(fmt.Sprintf("%s[%d]", prefix, i), e)
			}
		case nil:
			fmt.Fprintf(&sb, "%-32s: N/A\n", prefix)
		default:
			fmt.Fprintf(&sb, "%-32s: %v\n", prefix, t)
		}
	}
	walk("", generic)
	return sb.String() + time.Now().Format("# " + time.RFC3339)
}

func sortedKeys(m map[string]interface{}) []string {
	keys := make([]string, 0, len(m))
	for k := range m {
		keys = append(keys, k)
	}
	for i := 1; i < len(keys); i++ {
		for j := i; j > 0 && keys[j] < keys[j-1]; j-- {
			keys[j], keys[j-1] = keys[j-1], keys[j]
		}
	}
	return keys
}
