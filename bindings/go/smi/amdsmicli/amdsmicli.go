// Package amdsmicli is the test oracle: ground truth from the `amd-smi` command
// line tool in JSON form.  It plays the role of the reference's nvsmi package
// (bindings/go/nvml/nvsmi/nvsmi.go:17-43), which shells out to nvidia-smi in CSV form.
package amdsmicli

import (
	"bytes"
	"encoding/json"
	"fmt"
	"os/exec"
	"strconv"
	"strings"
)

const bin = "amd-smi"

// Available reports whether the tool can be run at all.
func Available() bool {
	_, err := exec.LookPath(bin)
	return err == nil
}

// Query runs `amd-smi <section> [-g gpu] --json` and returns one flattened map per
// GPU: nested objects become dotted keys, {"value": v, "unit": u} leaves become v,
// "N/A" becomes nil.
func Query(section string, gpu int) ([]map[string]interface{}, error) {
	args := []string{section}
	if gpu >= 0 {
		args = append(args, "-g", strconv.Itoa(gpu))
	}
	args = append(args, "--json")
	var stdout, stderr bytes.Buffer
	cmd := exec.Command(bin, args...)
	cmd.Stdout, cmd.Stderr = &stdout, &stderr
	if err := cmd.Run(); err != nil {
		return nil, fmt.Errorf("%s %s: %v: %s", bin, strings.Join(args, " "), err, stderr.String())
	}
	return Parse(stdout.Bytes())
}

// Parse accepts both output shapes of the tool: a bare list, or {"gpu_data": [...]}.
func Parse(out []byte) ([]map[string]interface{}, error) {
	start := bytes.IndexAny(out, "{[")
	if start < 0 {
		return nil, fmt.Errorf("no JSON in amd-smi output")
	}
	var doc interface{}
	dec := json.NewDecoder(bytes.NewReader(out[start:]))
	dec.UseNumber()
	if err := dec.Decode(&doc); err != nil {
		return nil, err
	}
	if m, ok := doc.(map[string]interface{}); ok {
		if gd, ok := m["gpu_data"]; ok {
			doc = gd
		} else {
			doc = []interface{}{m}
		}
	}
	list, ok := doc.([]interface{})
	if !ok {
		return nil, fmt.Errorf("unexpected amd-smi JSON shape")
	}
	res := make([]map[string]interface{}, 0, len(list))
	for _, g := range list {
		flat := map[string]interface{}{}
		flatten("", g, flat)
		res = append(res, flat)
	}
	return res, nil
}

func leaf(v interface{}) interface{} {
	switch x := v.(type) {
	case json.Number:
		if i, err := x.Int64(); err == nil {
			return i
		}
		f, _ := x.Float64()
		return f
	case string:
		s := strings.TrimSpace(x)
		if s == "" || strings.EqualFold(s, "N/A") {
			return nil
		}
		if i, err := strconv.ParseInt(s, 0, 64); err == nil {
			return i
		}
		if f, err := strconv.ParseFloat(s, 64); err == nil {
			return f
		}
		return s
	}
	return v
}

func flatten(prefix string, v interface{}, out map[string]interface{}) {
	switch x := v.(type) {
	case map[string]interface{}:
		if val, ok := x["value"]; ok && len(x) <= 2 {
			out[strings.TrimSuffix(prefix, ".")] = leaf(val)
			return
		}
		for k, c := range x {
			flatten(prefix+k+".", c, out)
		}
	case []interface{}:
		for i, c := range x {
			flatten(prefix+strconv.Itoa(i)+".", c, out)
		}
	default:
		out[strings.TrimSuffix(prefix, ".")] = leaf(v)
	}
}

// DeviceCount is the number of GPUs the tool lists.
func DeviceCount() (uint, error) {
	l, err := Query("list", -1)
	return uint(len(l)), err
}
